# -*- coding: utf-8 -*-
"""ORACLE tooling (build container only): what the REFERENCE returns in `scores` when lookahead_generation runs with
output_scores=True, return_dict_in_generate=True.

The reference appends ONE entry per verify step, `model_kwargs['next_tokens_scores']` (common/pretrained_model.py:1195, 1208-1209;
batch: pretrained_model_batch.py:1247, 1263-1264) — and that word is only written on steps WITHOUT drafts (:795; batch :807): the
prefill and any step whose retrieval came back empty.  On a step with drafts the tuple receives the previous no-draft step's tensor
again.  The vectors below pin exactly that: per step the scores row(s) the reference returned, plus which steps were no-draft steps.

Writes tests/golden/llama_tiny_scores_fp32.npz:
  bs1_*      single sequence (pretrained_model.py), two requests: plain greedy, and with RepetitionPenaltyLogitsProcessor(1.3)
  b3pad_*    batch of 3 left-padded prompts (pretrained_model_batch.py), one request, no processors
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden_batch import build_reference_batch_model, case_prompts  # noqa: E402
from oracle.gen_golden_model import OUT, build_reference_model, import_reference, tiny_prompt  # noqa: E402

PENALTY = 1.3


def _stack(scores):
    return np.stack([s.float().numpy() for s in scores], 0)          # [steps][bs][vocab]


def main():
    from transformers import LogitsProcessorList, MaxLengthCriteria, RepetitionPenaltyLogitsProcessor, StoppingCriteriaList
    LookaheadCache, _, LlamaForCausalLM = import_reference()
    save = {'penalty': np.array(PENALTY)}

    model = build_reference_model(LlamaForCausalLM, torch.float32)
    prompt = tiny_prompt()
    model.lookahead_cache = LookaheadCache()
    save['bs1_prompt'] = np.array(prompt)
    for r, procs in enumerate([LogitsProcessorList(), LogitsProcessorList([RepetitionPenaltyLogitsProcessor(PENALTY)])]):
        ids = torch.tensor([prompt], dtype=torch.long)
        dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12,
              'max_query_length': 2, 'stop_words': {}}
        with torch.no_grad():
            out = model.lookahead_generation(ids, logits_processor=procs,
                                             stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=len(prompt) + 64)]),
                                             pad_token_id=0, eos_token_id=2, return_dict_in_generate=True, output_scores=True,
                                             attention_mask=torch.ones_like(ids), decoding_kwargs=dk, use_cache=True)
        sc = _stack(out.scores)
        assert sc.shape[0] == len(out.kwargs['dls'])
        save[f'bs1_r{r}_sequences'] = np.array(out.sequences[0].tolist())
        save[f'bs1_r{r}_dls'] = np.array(out.kwargs['dls'])
        save[f'bs1_r{r}_edls'] = np.array(out.kwargs['edls'])
        save[f'bs1_r{r}_scores'] = sc[:, 0]
        fresh = [i for i in range(len(sc)) if i == 0 or not np.array_equal(sc[i], sc[i - 1])]
        print('bs1 run', r, 'steps', len(sc), 'dls', out.kwargs['dls'][:10], 'distinct score rows first seen at steps', fresh)

    bmodel = build_reference_batch_model(torch.float32)
    ids, am = case_prompts(3, [40, 33, 25])
    P = ids.shape[1]
    bmodel.lookahead_cache = LookaheadCache()
    dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12,
          'max_query_length': 2, 'stop_words': {}}
    with torch.no_grad():
        out = bmodel.lookahead_generation(torch.from_numpy(ids), logits_processor=LogitsProcessorList(),
                                          stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=P + 48)]),
                                          pad_token_id=0, eos_token_id=2, return_dict_in_generate=True, output_scores=True,
                                          attention_mask=torch.from_numpy(am), decoding_kwargs=dk, use_cache=True)
    sc = _stack(out.scores)
    save['b3pad_ids'], save['b3pad_am'] = ids, am
    save['b3pad_sequences'] = out.sequences.numpy().copy()
    save['b3pad_dls'] = np.array(out.kwargs['dls'])
    save['b3pad_edls'] = np.array(out.kwargs['edls'])
    save['b3pad_scores'] = sc
    print('b3pad steps', len(sc), 'dls', out.kwargs['dls'][:12])
    np.savez_compressed(os.path.join(OUT, 'llama_tiny_scores_fp32.npz'), **save)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
