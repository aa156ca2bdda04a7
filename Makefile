# convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly)
.PHONY: build test-cpu test-gpu bench golden

build:
	bash painlessinferenceacceleration_amd/csrc/build.sh

test-cpu: build
	python -m pytest tests -x -q -m "not gpu"

test-gpu: build
	python -m pytest tests -x -q -m gpu

bench: build
	python bench.py

# regenerate the golden vectors from the reference (needs /root/reference; build container only)
golden:
	python oracle/gen_golden.py && python oracle/gen_golden_model.py && python oracle/gen_golden_batch.py && \
	python oracle/gen_golden_moe.py && python oracle/gen_golden_processors.py && python oracle/gen_golden_mem.py && \
	python oracle/gen_golden_noisy.py && python oracle/gen_golden_batch_processors.py && python oracle/gen_golden_bench_trace.py && \
	python oracle/gen_golden_scores.py && python oracle/gen_golden_headdim.py
