# -*- coding: utf-8 -*-
"""API types of the lookahead path (mirror of lookahead/lookahead/common/lookahead_generation_utils.py:19-76)."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, Optional, Tuple


class GenerationMode(str, Enum):
    GREEDY_SEARCH = "greedy_search"
    LOOKAHEAD_GENERATION = "lookahead_generation"


class LookaheadGenerationConfig(object):
    """Mirror of LookaheadGenerationConfig(GenerationConfig) (lookahead_generation_utils.py:19-29): the lookahead fields with
    the reference's defaults, plus whatever generation fields the caller passes (max_new_tokens, eos_token_id, ...), which
    generate(generation_config=...) reads."""

    def __init__(self, **kwargs):
        self.use_lookahead = kwargs.pop('use_lookahead', False)
        self.debug_lookahead = kwargs.pop('debug_lookahead', False)
        self.decoding_length = kwargs.pop('decoding_length', 63)
        self.branch_length = kwargs.pop('branch_length', 12)
        self.decoding_mode = kwargs.pop('decoding_mode', 'hier')
        self.decoding_kwargs = kwargs.pop('decoding_kwargs', {})
        self.inputs_embeds_position = kwargs.pop('inputs_embeds_position', False)
        self.max_query_length = kwargs.pop('max_query_length', 2)
        self.stop_words = kwargs.pop('stop_words', None)
        self.tokenizer = kwargs.pop('tokenizer', None)
        for k, v in kwargs.items():            # max_new_tokens, max_length, eos_token_id, pad_token_id, do_sample, ...
            setattr(self, k, v)

    def to_decoding_kwargs(self) -> Dict[str, Any]:
        out = dict(self.decoding_kwargs or {})
        out.update({'use_lookahead': self.use_lookahead, 'debug_lookahead': self.debug_lookahead,
                    'decoding_mode': self.decoding_mode, 'decoding_length': self.decoding_length,
                    'branch_length': self.branch_length, 'max_query_length': self.max_query_length,
                    'stop_words': self.stop_words if self.stop_words is not None else {}, 'tokenizer': self.tokenizer})
        return out


@dataclass
class LookaheadDecoderOnlyOutput(object):
    """sequences + kwargs{dls, edls, fts, qts} (pretrained_model.py:1256-1266)."""
    sequences: Any = None
    scores: Optional[Tuple] = None
    attentions: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    kwargs: Dict[str, Any] = field(default_factory=dict)


# ---------------------------------------------------------------------------------------------------- generate() front door
_GEN_FIELDS = ('max_length', 'max_new_tokens', 'min_length', 'min_new_tokens', 'eos_token_id', 'pad_token_id', 'do_sample',
               'repetition_penalty', 'no_repeat_ngram_size', 'bad_words_ids', 'temperature', 'top_k', 'top_p', 'max_time',
               'return_dict_in_generate', 'output_scores')


@dataclass
class GenerateArgs(object):
    """What the reference's generate() derives before it dispatches to a decoding loop (pretrained_model.py:213-372)."""
    max_length: int = 0
    eos_token_id: Any = None
    pad_token_id: Any = None
    do_sample: bool = False
    return_dict_in_generate: bool = False
    output_scores: bool = False
    logits_processor: Any = None            # LogitsProcessorList: config-derived processors, then the caller's
    logits_warper: Any = None               # LogitsProcessorList of warpers (do_sample only)
    stopping_criteria: Any = None           # StoppingCriteriaList: MaxLengthCriteria (+ MaxTimeCriteria), then the caller's
    decoding_kwargs: Dict[str, Any] = field(default_factory=dict)


def _merge_lists(default_list, custom_list, what):
    """transformers' _merge_criteria_processor_list: the caller's objects go BEHIND the config-derived ones; passing an object of a
    type generate() already built from the configuration is an error there, and here."""
    if not custom_list:
        return default_list
    for d in default_list:
        for c in custom_list:
            if type(c) is type(d):
                raise ValueError(f'A custom {what} of type {type(c)} with values {c} has been passed to `generate`, but it has '
                                 f'already been created with the values {d}. {d} has been created by passing the corresponding '
                                 f'arguments to generate or by the model\'s config default values.')
    default_list.extend(custom_list)
    return default_list


def _on_device(cls, *args, device=None):
    """transformers >= 4.4x: the eos-aware processors keep their eos ids as a tensor created on `device` (default 'cpu') and compare
    it with the scores' vocabulary index — HF's _get_logits_processor passes device=input_ids.device; older releases take no such
    argument."""
    if device is not None:
        try:
            return cls(*args, device=device)
        except TypeError:
            pass
    return cls(*args)


def resolve_generate_args(model_generation_config, input_length, generation_config=None, logits_processor=None,
                          stopping_criteria=None, device=None, **kwargs):
    """The front half of the reference's generate() (common/pretrained_model.py:213-372) for the two modes this package serves.

    Precedence of every generation field: explicit keyword > `generation_config=` argument > the model's own generation_config
    (`generation_config.update(**kwargs)`, :229).  Then, in the reference's order:
      * pad_token_id defaults to the first eos id (:239-249);
      * max_length = max_new_tokens + prompt length when max_new_tokens is given (:325-336);
      * `_get_logits_processor` (:350-356): repetition_penalty, no_repeat_ngram_size, bad_words_ids, min_length, min_new_tokens —
        the transformers classes themselves, in transformers' order — then the caller's `logits_processor` list merged behind them;
      * `_get_stopping_criteria` (:359-361): MaxLengthCriteria, MaxTimeCriteria, then the caller's `stopping_criteria`;
      * warpers (temperature, top_k, top_p) are built for do_sample — the reference hands them to `sample()` only (:465-479); its
        LOOKAHEAD branch passes processors and criteria and NO warper (:428-441), and so does generate() here.
    `device`: where the logits rows handed to the processors live (the engine's device) — the eos-aware processors build their
    eos tensor there.
    Returns (GenerateArgs, leftover kwargs = model kwargs such as attention_mask)."""
    from transformers import (LogitsProcessorList, MaxLengthCriteria, MaxTimeCriteria, MinLengthLogitsProcessor,
                              MinNewTokensLengthLogitsProcessor, NoBadWordsLogitsProcessor, NoRepeatNGramLogitsProcessor,
                              RepetitionPenaltyLogitsProcessor, StoppingCriteriaList, TemperatureLogitsWarper, TopKLogitsWarper,
                              TopPLogitsWarper)
    val = {}
    for k in _GEN_FIELDS:
        if k in kwargs and kwargs[k] is not None:
            val[k] = kwargs.pop(k)
            continue
        kwargs.pop(k, None)
        v = getattr(generation_config, k, None) if generation_config is not None else None
        if v is None and k not in ('max_length', 'max_new_tokens'):          # the model's defaults never shorten an explicit budget
            v = getattr(model_generation_config, k, None) if model_generation_config is not None else None
        val[k] = v
    # decoding_kwargs: the keyword, else the config's (LookaheadGenerationConfig.to_decoding_kwargs / attributes, :232-233)
    dk = kwargs.pop('decoding_kwargs', None)
    if dk is None and generation_config is not None:
        if hasattr(generation_config, 'to_decoding_kwargs'):
            dk = generation_config.to_decoding_kwargs()
        else:
            dk = dict(getattr(generation_config, 'decoding_kwargs', {}) or {})
            for k in ('use_lookahead', 'debug_lookahead', 'decoding_length', 'branch_length', 'decoding_mode'):
                if hasattr(generation_config, k):
                    dk.setdefault(k, getattr(generation_config, k))
    dk = dict(dk or {})
    eos = val['eos_token_id']
    pad = val['pad_token_id']
    if pad is None and eos is not None:
        pad = eos[0] if isinstance(eos, (list, tuple)) else eos
    if val['max_new_tokens'] is not None:
        max_length = int(val['max_new_tokens']) + int(input_length)
    elif val['max_length'] is not None:
        max_length = int(val['max_length'])
    else:
        max_length = int(input_length) + 20
    eos_list = [eos] if isinstance(eos, int) else (list(eos) if eos is not None else None)
    procs = LogitsProcessorList()
    rp = val['repetition_penalty']
    if rp is not None and float(rp) != 1.0:
        procs.append(RepetitionPenaltyLogitsProcessor(penalty=float(rp)))
    if val['no_repeat_ngram_size'] is not None and int(val['no_repeat_ngram_size']) > 0:
        procs.append(NoRepeatNGramLogitsProcessor(int(val['no_repeat_ngram_size'])))
    if val['bad_words_ids'] is not None:
        procs.append(NoBadWordsLogitsProcessor(val['bad_words_ids'], eos_list))
    if val['min_length'] is not None and eos_list is not None and int(val['min_length']) > 0:
        procs.append(_on_device(MinLengthLogitsProcessor, int(val['min_length']), eos_list, device=device))
    if val['min_new_tokens'] is not None and eos_list is not None and int(val['min_new_tokens']) > 0:
        procs.append(_on_device(MinNewTokensLengthLogitsProcessor, int(input_length), int(val['min_new_tokens']), eos_list, device=device))
    procs = _merge_lists(procs, list(logits_processor) if logits_processor is not None else None, 'logits processor')
    warpers = LogitsProcessorList()
    if bool(val['do_sample']):
        if val['temperature'] is not None and float(val['temperature']) != 1.0:
            warpers.append(TemperatureLogitsWarper(float(val['temperature'])))
        if val['top_k'] is not None and int(val['top_k']) != 0:
            warpers.append(TopKLogitsWarper(top_k=int(val['top_k'])))
        if val['top_p'] is not None and float(val['top_p']) < 1.0:
            warpers.append(TopPLogitsWarper(top_p=float(val['top_p'])))
    crit = StoppingCriteriaList([MaxLengthCriteria(max_length=max_length)])
    if val['max_time'] is not None:
        crit.append(MaxTimeCriteria(max_time=float(val['max_time'])))
    crit = _merge_lists(crit, list(stopping_criteria) if stopping_criteria is not None else None, 'stopping criteria')
    ga = GenerateArgs(max_length=max_length, eos_token_id=eos, pad_token_id=pad, do_sample=bool(val['do_sample']),
                      return_dict_in_generate=bool(val['return_dict_in_generate']), output_scores=bool(val['output_scores']),
                      logits_processor=procs, logits_warper=warpers, stopping_criteria=crit, decoding_kwargs=dk)
    return ga, kwargs
