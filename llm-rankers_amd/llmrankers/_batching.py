"""Host-side batching that reproduces the reference's tokenise-all-then-pad-per-batch behaviour WITHOUT padding.

The reference tokenises every prompt at once with no truncation (ref: llmrankers/pairwise.py:17-26, EOS is
appended by the tokenizer), then a DataLoader + DataCollatorWithPadding(padding='longest') right-pads each
batch (ref: llmrankers/pointwise.py:90-101), forking 4 worker processes per rerank() call.  The engine takes
ragged token lists, so nothing is padded and nothing is forked; only the integer bookkeeping the reference
derives from the padded shape (total_prompt_tokens, ref: pointwise.py:107,114) is reproduced here.
"""
from typing import Iterator, List, Sequence, Tuple


def tokenize_prompts(tokenizer, prompts: Sequence[str]) -> List[List[int]]:
    if not prompts:
        return []
    return [list(ids) for ids in tokenizer(list(prompts))["input_ids"]]


def batches(n_items: int, batch_size: int) -> Iterator[Tuple[int, int]]:
    """[start, end) ranges in order; the last one may be short (drop_last=False, shuffle=False)."""
    for s in range(0, n_items, batch_size):
        yield s, min(s + batch_size, n_items)


def padded_token_count(seqs: Sequence[Sequence[int]]) -> int:
    """B x L_longest: what `input_ids.shape[0] * input_ids.shape[1]` is after the reference's collator."""
    return len(seqs) * max(len(s) for s in seqs) if seqs else 0
