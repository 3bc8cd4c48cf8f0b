"""Test-only runtime backed by the numpy oracle — lets the CPU suite exercise the rankers' host logic
(prompting, batching, counters, sort drivers, result assembly) against the reference goldens without a GPU.
The product never imports this (or anything under oracle/); production rankers build the HIP engine."""
import numpy as np

from oracle.t5_numpy import T5Oracle


class OracleRuntime:
    model_type = "t5"

    def __init__(self, dims, state):
        self.dims = dims
        self.orc = T5Oracle(dims, state)
        self.decoder_start_token_id = 0
        self.config = dims.to_hf_config()

    def score(self, seqs, dec_prefix, out_ids):
        return self.orc.score_last(seqs, dec_prefix, out_ids)

    def qlm(self, seqs, labels):
        return self.orc.qlm(seqs, labels)

    def greedy(self, seqs, dec_prefix, max_new, eos_id=1, pad_id=0):
        toks = self.orc.greedy(seqs, dec_prefix, max_new, eos_id, pad_id).astype(np.int32)
        # emulate "every row finished" early stop: columns after the last needed step are -1
        done_at = []
        for row in toks:
            hit = np.where(row == eos_id)[0]
            done_at.append(int(hit[0]) + 1 if len(hit) else max_new)
        steps = max(done_at)
        toks[:, steps:] = -1
        return toks


class OracleLlamaRuntime:
    """Llama counterpart: the numpy oracle behind LlamaRuntime's interface (greedy1 / last_logits)."""
    model_type = "llama"

    def __init__(self, dims, state):
        from oracle.llama_numpy import LlamaOracle
        self.dims, self.orc, self.config = dims, LlamaOracle(dims, state), dims.to_hf_config()

    def greedy1(self, seqs):
        return self.orc.greedy1(seqs)

    def last_logits(self, seqs, out_ids):
        return self.orc.last_logits(seqs, out_ids)
