"""DryEngine: a stand-in for llmrankers._engine.RkEngine that needs no GPU - `bench.py --dry_ranks` and the CPU tests use it to
walk the N-rank control flow of the bench (rank bring-up, communicator, sharding, one gather per launch sequence, the gather
checks, the JSON line) with N gloo processes before the first real 8-GPU run does.  Scores are a deterministic hash of each
sequence's tokens (batch-independent like the engine's), the 'device' buffers are numpy arrays and torch.distributed ('gloo')
carries what RCCL carries on the GPU box.  Test / tooling infrastructure: nothing in llmrankers imports this."""
import time

import numpy as np


def _gloo_all_gather_flat(local, width):
    """[world, width] float32 through ONE torch.distributed all_gather on gloo: what the engine's RCCL all_gather does on the GPU box."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size()
    local = np.asarray(local, dtype=np.float32).reshape(-1)
    buf = torch.zeros(width, dtype=torch.float32)
    buf[:len(local)] = torch.as_tensor(local)
    out = torch.empty(ws * width, dtype=torch.float32)
    dist.all_gather_into_tensor(out, buf)
    return out.numpy().reshape(ws, width)


def fake_logits(seqs, out_ids):
    out = np.empty((len(seqs), len(out_ids)), np.float32)
    for i, s in enumerate(seqs):
        h = (int(np.asarray(s, dtype=np.int64).sum()) * 2654435761 + len(s) * 40503) & 0xFFFFFFFF
        for j, o in enumerate(out_ids):
            out[i, j] = np.float32(((h ^ (int(o) * 97)) % 100003) / 100003.0 * 8.0 - 4.0)
    return out


class DryEngine:
    num_slots = 2
    h = True

    def __init__(self, dims, device=0, max_tokens=16384, max_seqs=128, max_dec_len=4):
        from types import SimpleNamespace
        self.dims, self.device = dims, device
        self.desc = SimpleNamespace(max_tokens=max_tokens, max_seqs=max_seqs, max_dec_len=max_dec_len)
        self.comm_rank, self.comm_world, self.comm_capacity = 0, 1, 0
        self._staged, self._scores, self._gathered = {}, {}, {}
        self._send, self._t0 = None, 0.0
        self.calls = {"score": 0, "gather": 0, "append": 0, "append_host": 0, "init": 0}

    def load_state(self, tensors):
        for _ in tensors:
            pass
        return self

    def set_option(self, key, value):
        pass

    def close(self):
        self.h = None

    # -- scoring -------------------------------------------------------------------------------------------------
    def score(self, seqs, dec_prefix, out_ids):
        assert 0 < len(seqs) <= self.desc.max_seqs and sum(len(s) for s in seqs) <= self.desc.max_tokens
        self.calls["score"] += 1
        self._scores[0] = fake_logits(seqs, out_ids)
        return self._scores[0].copy()

    def qlm(self, seqs, labels):
        self.calls["score"] += 1
        self._scores[0] = fake_logits(seqs, [len(labels)])
        return self._scores[0].reshape(-1).copy()

    def stage(self, seqs, slot=0):
        assert 0 < len(seqs) <= self.desc.max_seqs and sum(len(s) for s in seqs) <= self.desc.max_tokens
        self._staged[slot] = [list(s) for s in seqs]

    def score_staged(self, dec_prefix, out_ids, slot=0):
        self.calls["score"] += 1
        self._scores[slot] = fake_logits(self._staged[slot], out_ids)

    def read_scores(self, slot=0):
        return self._scores[slot].copy()

    def sync(self):
        pass

    def timer_begin(self):
        self._t0 = time.perf_counter()

    def timer_end(self):
        return (time.perf_counter() - self._t0) * 1e3

    # -- communicator (gloo instead of RCCL) ---------------------------------------------------------------------------
    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world, cap):
        assert uid == bytes(range(128)), "the id did not travel from rank 0"
        self.calls["init"] += 1
        self.comm_rank, self.comm_world, self.comm_capacity = rank, world, cap
        self._send = np.zeros(cap, np.float32)

    def comm_library_info(self):
        return "dry-run (gloo)|0"

    def comm_all_gather(self, n_floats, slot=0):
        assert 0 < n_floats <= self.comm_capacity
        self.calls["gather"] += 1
        flat = self._scores[slot].reshape(-1)
        self._gathered[slot] = _gloo_all_gather_flat(flat[:n_floats], n_floats)

    def comm_read_gathered(self, slot=0):
        return self._gathered[slot].copy()

    def comm_append(self, n, offset, slot=0):
        self.calls["append"] += 1
        self._send[offset:offset + n] = self._scores[slot].reshape(-1)[:n]

    def comm_append_host(self, values, offset):
        v = np.asarray(values, dtype=np.float32).reshape(-1)
        self.calls["append_host"] += 1
        self._send[offset:offset + len(v)] = v

    def comm_all_gather_appended(self, n):
        assert 0 < n <= self.comm_capacity
        self.calls["gather"] += 1
        return _gloo_all_gather_flat(self._send[:n], n)

    def comm_destroy(self):
        self.comm_world = 1
