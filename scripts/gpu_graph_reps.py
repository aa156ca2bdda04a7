import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""What does a graph launch cost beyond its kernels?  The Llama-2-7B verify step captured n times into ONE graph
(la_debug_set key 11; same input block each repetition, only the last one publishes): wall time per launch / n against n.

    python scripts/gpu_graph_reps.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import fixed_t64b8_tree                                        # noqa: E402
from painlessinferenceacceleration_amd._lib import check, lib            # noqa: E402
from painlessinferenceacceleration_amd.llama_engine import LlamaShape, LlamaVerifyEngine, random_weights   # noqa: E402


def main():
    torch.cuda.set_device(0)
    shape = LlamaShape.llama2_7b()
    eng = LlamaVerifyEngine(shape, random_weights(shape, seed=0, device='cuda:0', decisive=True), max_length=4096, consume_state_dict=True)
    rs = np.random.RandomState(0)
    prompt = rs.randint(3, shape.vocab, size=512).tolist()
    _, _, rows = fixed_t64b8_tree()
    ids = rs.randint(3, shape.vocab, size=64).astype(np.int32)
    out = []
    for reps in (1, 2, 4, 8, 1):
        check(lib.la_lab_set(11, 1), 'debug_set')
        eng.reset()
        ids[0] = eng.prefill(prompt, fast=False)
        check(lib.la_lab_set(11, reps), 'debug_set')        # the prompt went through the plain graph: same context for every n
        for _ in range(3):
            eng.step(ids, rows, mode=2)
        torch.cuda.synchronize()
        n = 24 // reps + 2
        t0 = time.perf_counter()
        for _ in range(n):
            eng.step(ids, rows, mode=2)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        rec = {'steps_per_graph': reps, 'ms_per_launch': round(ms, 4), 'ms_per_step': round(ms / reps, 4), 'context_at_end': eng.n_keys}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    check(lib.la_lab_set(11, 1), 'debug_set')


if __name__ == '__main__':
    main()
