# -*- coding: utf-8 -*-
"""ORACLE tooling (build container only): the REFERENCE's batch lookahead_generation (bs > 1) with a non-empty
logits-processor list (RepetitionPenaltyLogitsProcessor) — the per-sample sequential accept walk of
common/pretrained_model_batch.py:814-931 with the processors applied to input_ids[b, :cur+i+2] (pads included) and the
batch-wise processor call of the prefill (:783).  Writes tests/golden/llama_tiny_batch_fp32_rep.npz: for each case the padded
prompts, masks, final sequences, dls / edls of two consecutive requests (the second one on the trie warmed by the first)."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden_batch import build_reference_batch_model, case_prompts  # noqa: E402
from oracle.gen_golden_model import OUT, import_reference  # noqa: E402

PENALTY = 1.3
# (name, batch size, valid prompt lengths (left-padded to the longest), decoding_length, max_new)
CASES = [
    ('b2', 2, [40, 40], 64, 48),
    ('b3pad', 3, [40, 33, 25], 64, 48),
    ('b3pad256', 3, [40, 33, 25], 256, 40),      # more than 64 draft rows per step: one block per sample on the product path
]


def main():
    from transformers import LogitsProcessorList, MaxLengthCriteria, RepetitionPenaltyLogitsProcessor, StoppingCriteriaList
    LookaheadCache = import_reference()[0]
    model = build_reference_batch_model(torch.float32)
    save = {'cases': np.array([c[0] for c in CASES]), 'penalty': np.array(PENALTY)}
    for name, bs, lengths, dl, max_new in CASES:
        ids, am = case_prompts(bs, lengths)
        P = ids.shape[1]
        model.lookahead_cache = LookaheadCache()
        save[f'{name}_ids'] = ids
        save[f'{name}_am'] = am
        save[f'{name}_cfg'] = np.array([bs, dl, max_new])
        for r in range(2):
            dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': dl, 'branch_length': 12,
                  'max_query_length': 2, 'stop_words': {}}
            with torch.no_grad():
                out = model.lookahead_generation(
                    torch.from_numpy(ids), logits_processor=LogitsProcessorList([RepetitionPenaltyLogitsProcessor(PENALTY)]),
                    stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=P + max_new)]),
                    pad_token_id=0, eos_token_id=2, return_dict_in_generate=True,
                    attention_mask=torch.from_numpy(am), decoding_kwargs=dk, use_cache=True)
            save[f'{name}_r{r}_sequences'] = out.sequences.numpy().copy()
            save[f'{name}_r{r}_dls'] = np.array(out.kwargs['dls'])
            save[f'{name}_r{r}_edls'] = np.array(out.kwargs['edls'])
            print(name, 'run', r, 'dls', out.kwargs['dls'][:14], 'edls', out.kwargs['edls'][:14])
    np.savez_compressed(os.path.join(OUT, 'llama_tiny_batch_fp32_rep.npz'), **save)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
