# -*- coding: utf-8 -*-
"""API types of the lookahead path (mirror of lookahead/lookahead/common/lookahead_generation_utils.py:19-76)."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, Optional, Tuple


class GenerationMode(str, Enum):
    GREEDY_SEARCH = "greedy_search"
    LOOKAHEAD_GENERATION = "lookahead_generation"


@dataclass
class LookaheadGenerationConfig(object):
    """The decoding_kwargs dict as a typed object (defaults = pretrained_model.py:674-680)."""
    use_lookahead: bool = True
    debug_lookahead: bool = False
    decoding_mode: str = 'hier'
    decoding_length: int = 64
    branch_length: int = 12
    max_query_length: int = 2
    stop_words: Optional[dict] = None
    tokenizer: Any = None

    def to_decoding_kwargs(self) -> Dict[str, Any]:
        return {'use_lookahead': self.use_lookahead, 'debug_lookahead': self.debug_lookahead,
                'decoding_mode': self.decoding_mode, 'decoding_length': self.decoding_length,
                'branch_length': self.branch_length, 'max_query_length': self.max_query_length,
                'stop_words': self.stop_words if self.stop_words is not None else {}, 'tokenizer': self.tokenizer}


@dataclass
class LookaheadDecoderOnlyOutput(object):
    """sequences + kwargs{dls, edls, fts, qts} (pretrained_model.py:1256-1266)."""
    sequences: Any = None
    scores: Optional[Tuple] = None
    attentions: Optional[Tuple] = None
    hidden_states: Optional[Tuple] = None
    kwargs: Dict[str, Any] = field(default_factory=dict)
