"""Test-only runtime backed by the numpy oracle — lets the CPU suite exercise the rankers' host logic
(prompting, batching, counters, sort drivers, result assembly) against the reference goldens without a GPU.
The product never imports this (or anything under oracle/); production rankers build the HIP engine."""
import numpy as np

from oracle.t5_numpy import T5Oracle


def gloo_all_gather_flat(local, width):
    """[world, width] float32: every rank's `local` (<= width values, zero padded) through ONE torch.distributed all_gather of
    the process group's own backend (the CPU tests run on gloo) - the stand-in for the engine's RCCL all_gather, which the real
    runtime issues itself (rk_comm_all_gather_slot / rk_comm_all_gather_appended)."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size()
    local = np.asarray(local, dtype=np.float32).reshape(-1)
    buf = torch.zeros(width, dtype=torch.float32)
    buf[:len(local)] = torch.as_tensor(local)
    out = torch.empty(ws * width, dtype=torch.float32)
    dist.all_gather_into_tensor(out, buf)
    return out.numpy().reshape(ws, width)


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

    def host_all_gather(self, local, width):
        """candidate sharding without an engine communicator (PointwiseLlmRanker._rerank_sharded): the gloo stand-in"""
        return gloo_all_gather_flat(local, width)

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


class FakeCommEngine:
    """Stands in for RkEngine behind the REAL T5Runtime (T5Runtime.from_engine) in the multi-process CPU tests: the oracle
    computes the scores, a numpy array plays the device send buffer of rk_comm_append_scores_slot, and gloo carries what
    RCCL carries on the GPU box.  Capacities are tiny on purpose, so that a rank's share of a query takes several calls."""

    def __init__(self, dims, state, max_seqs=3, max_tokens=100000):
        from types import SimpleNamespace
        self.dims, self.orc = dims, T5Oracle(dims, state)
        self.desc = SimpleNamespace(max_tokens=max_tokens, max_seqs=max_seqs)
        self.comm_rank, self.comm_world = 0, 1
        self.calls = {"score": 0, "qlm": 0, "append": 0, "gather": 0, "init": 0}
        self._last = np.zeros(0, np.float32)
        self._send = None

    def score(self, seqs, dec_prefix, out_ids):
        assert 0 < len(seqs) <= self.desc.max_seqs
        self.calls["score"] += 1
        out = np.asarray(self.orc.score_last(seqs, dec_prefix, out_ids), dtype=np.float32)
        self._last = out.reshape(-1).copy()          # "slot 0's device score buffer" is overwritten by every call
        return out

    def qlm(self, seqs, labels):
        assert 0 < len(seqs) <= self.desc.max_seqs
        self.calls["qlm"] += 1
        out = np.asarray(self.orc.qlm(seqs, labels), dtype=np.float32)
        self._last = out.reshape(-1).copy()
        return out

    # staged form used by T5Runtime.score_batches (single-process pipelining): one slot, evaluated at read time
    num_slots = 1

    def stage(self, seqs, slot=0):
        self._staged = [list(x) for x in seqs]

    def score_staged(self, dec_prefix, out_ids, slot=0):
        self._staged_args = (list(dec_prefix), list(out_ids))

    def read_scores(self, slot=0):
        return self.score(self._staged, *self._staged_args)

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world, cap):
        assert uid == bytes(range(128)), "the id did not travel from rank 0"
        self.calls["init"] += 1
        self.comm_rank, self.comm_world = rank, world
        self.comm_capacity = cap
        self._send = np.full(cap, np.nan, np.float32)

    def comm_append_host(self, values, offset):
        v = np.asarray(values, dtype=np.float32).reshape(-1)
        assert offset + len(v) <= len(self._send)
        self.calls["append_host"] = self.calls.get("append_host", 0) + 1
        self._send[offset:offset + len(v)] = v

    def comm_append(self, n, offset, slot=0):
        assert slot == 0 and n <= len(self._last)
        self.calls["append"] += 1
        self._send[offset:offset + n] = self._last[:n]

    def comm_all_gather_appended(self, n):
        self.calls["gather"] += 1
        return gloo_all_gather_flat(np.nan_to_num(self._send[:n], nan=-777.0), n)
