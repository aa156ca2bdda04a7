# -*- coding: utf-8 -*-
"""ORACLE tooling (build container only): the REFERENCE's lookahead_generation with a non-empty logits-processor list
(RepetitionPenaltyLogitsProcessor) — the sequential accept path of pretrained_model.py:825-875.
Writes tests/golden/llama_tiny_fp32_rep.npz (sequences / dls / edls of two consecutive requests)."""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden_model import OUT, build_reference_model, import_reference, tiny_prompt  # noqa: E402

PENALTY = 1.3


def main():
    from transformers import LogitsProcessorList, MaxLengthCriteria, RepetitionPenaltyLogitsProcessor, StoppingCriteriaList
    LookaheadCache, _, LlamaForCausalLM = import_reference()
    model = build_reference_model(LlamaForCausalLM, torch.float32)
    prompt = tiny_prompt()
    model.lookahead_cache = LookaheadCache()
    save = {'prompt': np.array(prompt), 'penalty': np.array(PENALTY)}
    for r in range(2):
        ids = torch.tensor([prompt], dtype=torch.long)
        dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12,
              'max_query_length': 2, 'stop_words': {}}
        with torch.no_grad():
            out = model.lookahead_generation(ids, logits_processor=LogitsProcessorList([RepetitionPenaltyLogitsProcessor(PENALTY)]),
                                             stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=len(prompt) + 64)]),
                                             pad_token_id=0, eos_token_id=2, return_dict_in_generate=True,
                                             attention_mask=torch.ones_like(ids), decoding_kwargs=dk, use_cache=True)
        save[f'r{r}_sequences'] = np.array(out.sequences[0].tolist())
        save[f'r{r}_dls'] = np.array(out.kwargs['dls'])
        save[f'r{r}_edls'] = np.array(out.kwargs['edls'])
        print('run', r, 'dls', out.kwargs['dls'][:12], 'edls', out.kwargs['edls'][:12])
    np.savez_compressed(os.path.join(OUT, 'llama_tiny_fp32_rep.npz'), **save)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
