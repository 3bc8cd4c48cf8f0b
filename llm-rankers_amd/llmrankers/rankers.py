"""Public result record and ranker base class — the outer drop-in surface.

Mirrors ref: llmrankers/rankers.py:5-17 (field names, positional order and method names are the contract
`run.py` and library users rely on: ref run.py:176,192-195; README.md:38-54).
"""
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class SearchResult:
    """One candidate of a first-stage ranking.  `text` is None in setwise results (ref: setwise.py:306,310)."""
    docid: str
    score: float
    text: Optional[str]


class LlmRanker:
    """Interface every ranker implements: rerank a candidate list for a query; truncate text by tokens."""

    def rerank(self, query: str, ranking: List[SearchResult]) -> List[SearchResult]:
        raise NotImplementedError

    def truncate(self, text: str, length: int) -> str:
        raise NotImplementedError
