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
