"""ctypes binding of librk_engine.so (C ABI: include/rk_engine.h).

north_star asks for a "thin C-ABI cffi layer"; cffi is not installed in this image,
so the same extern "C" surface is bound with the stdlib's ctypes.  There is no fallback: if the shared
library is missing or no gfx950 device is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "librk_engine.so")

RK_F32, RK_F16, RK_BF16 = 0, 1, 2


class RkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rk_engine error {code}: {msg}")
        self.code = code


class RkModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "d_model", "n_heads", "d_kv", "d_ff", "n_enc_layers",
                                         "n_dec_layers", "n_buckets", "max_distance", "gated_gelu", "tied_head")] + \
               [("eps", C.c_float)] + [(n, C.c_int32) for n in ("max_tokens", "max_seqs", "max_dec_len")]


class RkLlamaDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "hidden", "n_heads", "n_kv_heads", "head_dim", "intermediate", "n_layers",
                                         "tied_head")] + [("eps", C.c_float), ("rope_theta", C.c_float)] + \
               [(n, C.c_int32) for n in ("max_tokens", "max_seqs")]


# name -> (restype, argtypes); this table is also what tests check against include/rk_engine.h
_P = C.POINTER
_i32p, _f32p = _P(C.c_int32), _P(C.c_float)
ABI = {
    "rk_engine_create": (C.c_int, [_P(RkModelDesc), C.c_int, _P(C.c_void_p)]),
    "rk_engine_destroy": (None, [C.c_void_p]),
    "rk_last_error": (C.c_char_p, [C.c_void_p]),
    "rk_engine_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, _P(C.c_int64), C.c_int]),
    "rk_engine_finalize": (C.c_int, [C.c_void_p]),
    "rk_t5_score": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, _f32p]),
    "rk_t5_qlm": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, _i32p, C.c_int, _f32p]),
    "rk_t5_greedy": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p]),
    "rk_t5_greedy2": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, _i32p, C.c_int, _i32p, C.c_int, C.c_int, C.c_int, _i32p, _i32p]),
    "rk_t5_stage": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int]),
    "rk_t5_score_staged": (C.c_int, [C.c_void_p, _i32p, C.c_int, _i32p, C.c_int]),
    "rk_engine_sync": (C.c_int, [C.c_void_p]),
    "rk_engine_num_slots": (C.c_int, []),
    "rk_t5_stage_slot": (C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p, C.c_int]),
    "rk_t5_score_slot": (C.c_int, [C.c_void_p, C.c_int, _i32p, C.c_int, _i32p, C.c_int]),
    "rk_t5_read_scores_slot": (C.c_int, [C.c_void_p, C.c_int, _f32p, C.c_int]),
    "rk_t5_read_scores": (C.c_int, [C.c_void_p, _f32p, C.c_int]),
    "rk_t5_scores_device_ptr": (C.c_int, [C.c_void_p, _P(C.c_void_p)]),
    "rk_llama_create": (C.c_int, [_P(RkLlamaDesc), C.c_int, _P(C.c_void_p)]),
    "rk_llama_set_rope_scaling": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int]),
    "rk_llama_greedy1": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, _i32p]),
    "rk_llama_last_logits": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, _i32p, C.c_int, _f32p]),
    "rk_comm_unique_id": (C.c_int, [_P(C.c_uint8), C.c_int]),
    "rk_comm_init": (C.c_int, [C.c_void_p, _P(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int]),
    "rk_comm_world": (C.c_int, [C.c_void_p, _i32p, _i32p]),
    "rk_comm_capacity": (C.c_int, [C.c_void_p]),
    "rk_comm_library_info": (C.c_int, [C.c_char_p, C.c_int]),
    "rk_comm_append_host": (C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int]),
    "rk_comm_all_gather_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rk_comm_read_gathered_slot": (C.c_int, [C.c_void_p, C.c_int, _f32p, C.c_int]),
    "rk_comm_append_scores_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rk_comm_all_gather_appended": (C.c_int, [C.c_void_p, C.c_int]),
    "rk_comm_read_appended": (C.c_int, [C.c_void_p, _f32p, C.c_int]),
    "rk_comm_destroy": (C.c_int, [C.c_void_p]),
    "rk_timer_begin": (C.c_int, [C.c_void_p]),
    "rk_timer_end": (C.c_int, [C.c_void_p, _f32p]),
    "rk_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rk_profile_reset": (C.c_int, [C.c_void_p]),
    "rk_profile_num_classes": (C.c_int, []),
    "rk_profile_class_name": (C.c_char_p, [C.c_int]),
    "rk_profile_get": (C.c_int, [C.c_void_p, C.c_int, _P(C.c_double), _P(C.c_int64), _P(C.c_double), _P(C.c_double)]),
    "rk_engine_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "rk_abi_version": (C.c_int, []),
    "rk_rel_bucket": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rk_debug_gemm": (C.c_int, [C.c_void_p, _P(C.c_uint16), _P(C.c_uint16), _f32p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rk_debug_gemm_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    "rk_debug_read": (C.c_int64, [C.c_void_p, C.c_char_p, _f32p, C.c_int64]),
}

_lib = None


def load_library(path: Optional[str] = None, make_default: bool = False):
    """dlopen librk_engine.so and attach prototypes. Raises if the in-tree build is missing (no fallback).
    The product always loads the in-tree library (LIB_PATH).  `path` + make_default=True is for tools/ only: separately compiled
    A/B or measurement builds (tools/_lib.py); no environment variable redirects the product's library."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  If torch gets imported AFTER this library (e.g. via
    # transformers' tokenizer), the process would hold two HIP runtimes; importing torch first lets the loader
    # resolve our DT_NEEDED libamdhip64.so.7 to the copy already mapped.  Plumbing only — nothing here uses torch.
    if os.environ.get("RK_IMPORT_TORCH_FIRST", "1") == "1":
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU/PyTorch fallback for the hot path.")
    lib = C.CDLL(p)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)     # AttributeError here = the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if path is None or make_default:
        _lib = lib
    return lib


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def pack_ragged(seqs: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    """list of token-id sequences -> (tokens[T] int32, seq_offsets[B+1] int32): the engine's input boundary."""
    lens = [len(s) for s in seqs]
    off = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(lens, out=off[1:])
    tok = np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs]) if seqs else np.zeros(0, np.int32)
    return np.ascontiguousarray(tok, dtype=np.int32), off


class RkEngine:
    """One engine = one MI355X.  Thin object wrapper; all compute happens in the HIP library."""

    def __init__(self, dims, device: int = 0, max_tokens: int = 16384, max_seqs: int = 128, max_dec_len: int = 136):
        self.lib = load_library()
        self.dims = dims
        self.desc = RkModelDesc(vocab=dims.vocab, d_model=dims.d_model, n_heads=dims.n_heads, d_kv=dims.d_kv,
                                d_ff=dims.d_ff, n_enc_layers=dims.n_enc, n_dec_layers=dims.n_dec,
                                n_buckets=dims.n_buckets, max_distance=dims.max_distance,
                                gated_gelu=int(dims.gated), tied_head=int(dims.tied_head), eps=dims.eps,
                                max_tokens=max_tokens, max_seqs=max_seqs, max_dec_len=max_dec_len)
        h = C.c_void_p()
        rc = self.lib.rk_engine_create(C.byref(self.desc), device, C.byref(h))
        if rc != 0:
            raise RkError(rc, (self.lib.rk_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        self.comm_rank, self.comm_world = 0, 1

    # -- plumbing ------------------------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != 0:
            raise RkError(rc, (self.lib.rk_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rk_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------------------
    def load_tensor(self, name: str, arr: np.ndarray):
        if arr.dtype == np.float16:
            dt = RK_F16
        elif arr.dtype == np.float32:
            dt = RK_F32
        elif arr.dtype == np.uint16:      # raw bf16 bits (safetensors bf16 viewed as uint16)
            dt = RK_BF16
        else:
            arr, dt = arr.astype(np.float32), RK_F32
        arr = np.ascontiguousarray(arr)
        shape = (C.c_int64 * arr.ndim)(*arr.shape)
        self._chk(self.lib.rk_engine_load_tensor(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), dt, shape, arr.ndim))

    def load_state(self, tensors: Iterable[Tuple[str, np.ndarray]]):
        for name, arr in tensors:
            self.load_tensor(name, arr)
        self._chk(self.lib.rk_engine_finalize(self.h))
        return self

    # -- the three call shapes of the hot path ----------------------------------------------------------
    def score(self, seqs: Sequence[Sequence[int]], dec_prefix: Sequence[int], out_ids: Sequence[int]) -> np.ndarray:
        tok, off = pack_ragged(seqs)
        dp, oi = _i32(dec_prefix), _i32(out_ids)
        out = np.empty((len(seqs), len(oi)), dtype=np.float32)
        self._chk(self.lib.rk_t5_score(self.h, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs),
                                       dp.ctypes.data_as(_i32p), len(dp), oi.ctypes.data_as(_i32p), len(oi),
                                       out.ctypes.data_as(_f32p)))
        return out

    def qlm(self, seqs: Sequence[Sequence[int]], labels: Sequence[int]) -> np.ndarray:
        tok, off = pack_ragged(seqs)
        lab = _i32(labels)
        out = np.empty(len(seqs), dtype=np.float32)
        self._chk(self.lib.rk_t5_qlm(self.h, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs),
                                     lab.ctypes.data_as(_i32p), len(lab), out.ctypes.data_as(_f32p)))
        return out

    def greedy(self, seqs: Sequence[Sequence[int]], dec_prefix: Sequence[int], max_new: int, eos_id: int = 1,
               pad_id: int = 0, candidates: Optional[Sequence[int]] = None) -> Tuple[np.ndarray, int]:
        """`candidates` (max_new == 2 only): token ids the first new token is expected to be among - same result, one
        decoder pass instead of two (rk_t5_greedy2)."""
        tok, off = pack_ragged(seqs)
        dp = _i32(dec_prefix)
        out = np.empty((len(seqs), max_new), dtype=np.int32)
        steps = C.c_int32(0)
        if candidates is not None and len(candidates) and max_new == 2:
            cd = _i32(candidates)
            self._chk(self.lib.rk_t5_greedy2(self.h, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs),
                                             dp.ctypes.data_as(_i32p), len(dp), cd.ctypes.data_as(_i32p), len(cd), eos_id, pad_id,
                                             out.ctypes.data_as(_i32p), C.byref(steps)))
            return out, int(steps.value)
        self._chk(self.lib.rk_t5_greedy(self.h, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs),
                                        dp.ctypes.data_as(_i32p), len(dp), max_new, eos_id, pad_id,
                                        out.ctypes.data_as(_i32p), C.byref(steps)))
        return out, int(steps.value)

    # -- staged / async form (bench, multi-GPU) --------------------------------------------------------
    @property
    def num_slots(self) -> int:
        return int(self.lib.rk_engine_num_slots())

    def stage(self, seqs: Sequence[Sequence[int]], slot: int = 0):
        tok, off = pack_ragged(seqs)
        self._chk(self.lib.rk_t5_stage_slot(self.h, slot, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs)))
        self._slot_shape = getattr(self, "_slot_shape", {})
        self._slot_shape[slot] = [len(seqs), 0]

    def score_staged(self, dec_prefix: Sequence[int], out_ids: Sequence[int], slot: int = 0):
        """Enqueue encoder (encoder stream) + decoder/head (decoder stream) for the slot's batch; returns at once."""
        dp, oi = _i32(dec_prefix), _i32(out_ids)
        self._slot_shape[slot][1] = len(oi)
        self._chk(self.lib.rk_t5_score_slot(self.h, slot, dp.ctypes.data_as(_i32p), len(dp), oi.ctypes.data_as(_i32p), len(oi)))

    def sync(self):
        self._chk(self.lib.rk_engine_sync(self.h))

    def read_scores(self, slot: int = 0) -> np.ndarray:
        n, k = self._slot_shape[slot]
        out = np.empty((n, k), dtype=np.float32)
        self._chk(self.lib.rk_t5_read_scores_slot(self.h, slot, out.ctypes.data_as(_f32p), out.size))
        return out

    def scores_device_ptr(self) -> int:
        p = C.c_void_p()
        self._chk(self.lib.rk_t5_scores_device_ptr(self.h, C.byref(p)))
        return int(p.value)

    # -- multi-GPU score collection (RCCL inside the engine) -----------------------------------------------
    COMM_ID_BYTES = 128

    def comm_unique_id(self) -> bytes:
        """rank 0: the 128-byte RCCL id every rank needs for comm_init (ship it with any host-side channel)."""
        buf = (C.c_uint8 * self.COMM_ID_BYTES)()
        rc = self.lib.rk_comm_unique_id(buf, self.COMM_ID_BYTES)
        if rc != 0:
            raise RkError(rc, (self.lib.rk_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int, max_floats_per_rank: int):
        """Collective over all ranks (one engine = one process = one GPU)."""
        buf = (C.c_uint8 * self.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._chk(self.lib.rk_comm_init(self.h, buf, self.COMM_ID_BYTES, rank, world, max_floats_per_rank))
        self.comm_rank, self.comm_world = rank, world
        self.comm_capacity = int(self.lib.rk_comm_capacity(self.h))   # what every rank checks before it enters a gather

    def comm_library_info(self) -> str:
        """'<path of the RCCL library this process bound>|<ncclGetVersion code>' (the torch wheel bundles its own librccl;
        which one served the run belongs in the logs)."""
        buf = C.create_string_buffer(1024)
        n = self.lib.rk_comm_library_info(buf, 1024)
        if n < 0:
            raise RkError(n, (self.lib.rk_last_error(None) or b"").decode())
        return buf.value.decode()

    def comm_append_host(self, values, offset: int):
        """Put host floats at `offset` of the engine's send buffer (side data that travels with the scores in the one gather)."""
        v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
        self._chk(self.lib.rk_comm_append_host(self.h, v.ctypes.data_as(_f32p), v.size, offset))

    def comm_all_gather(self, n_floats: int, slot: int = 0):
        """Enqueue ONE RCCL all_gather of the slot's device score buffer behind the work that fills it (no sync)."""
        self._chk(self.lib.rk_comm_all_gather_slot(self.h, slot, n_floats))
        self._gather_n = getattr(self, "_gather_n", {})
        self._gather_n[slot] = n_floats

    def comm_read_gathered(self, slot: int = 0) -> np.ndarray:
        """[world, n_floats] float32 of the slot's last gather (waits for it)."""
        n = self._gather_n[slot]
        out = np.empty((self.comm_world, n), dtype=np.float32)
        self._chk(self.lib.rk_comm_read_gathered_slot(self.h, slot, out.ctypes.data_as(_f32p), out.size))
        return out

    def comm_append(self, n_floats: int, offset: int, slot: int = 0):
        """Copy the slot's last n_floats scores to `offset` of the engine's send buffer (device to device, no sync)."""
        self._chk(self.lib.rk_comm_append_scores_slot(self.h, slot, n_floats, offset))

    def comm_all_gather_appended(self, n_floats: int) -> np.ndarray:
        """ONE RCCL all_gather of the first n_floats of the send buffer -> [world, n_floats] float32 (waits for it)."""
        self._chk(self.lib.rk_comm_all_gather_appended(self.h, n_floats))
        out = np.empty((self.comm_world, n_floats), dtype=np.float32)
        self._chk(self.lib.rk_comm_read_appended(self.h, out.ctypes.data_as(_f32p), out.size))
        return out

    def comm_destroy(self):
        self._chk(self.lib.rk_comm_destroy(self.h))
        self.comm_rank, self.comm_world, self.comm_capacity = 0, 1, 0

    # -- measurement -------------------------------------------------------------------------------------
    def timer_begin(self):
        self._chk(self.lib.rk_timer_begin(self.h))

    def timer_end(self) -> float:
        ms = C.c_float(0)
        self._chk(self.lib.rk_timer_end(self.h, C.byref(ms)))
        return float(ms.value)

    def profile(self, on: bool):
        self._chk(self.lib.rk_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._chk(self.lib.rk_profile_reset(self.h))

    def profile_report(self) -> dict:
        rep = {}
        for c in range(self.lib.rk_profile_num_classes()):
            ms, n, fl, by = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
            self._chk(self.lib.rk_profile_get(self.h, c, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            rep[self.lib.rk_profile_class_name(c).decode()] = {"ms": ms.value, "launches": n.value, "flops": fl.value,
                                                              "bytes": by.value}
        return rep

    def set_option(self, key: str, value: int):
        self._chk(self.lib.rk_engine_set_option(self.h, key.encode(), int(value)))

    # -- debug ---------------------------------------------------------------------------------------------
    def debug_gemm(self, a16: np.ndarray, w16: np.ndarray, use_glds=True) -> np.ndarray:
        a16 = np.ascontiguousarray(a16, dtype=np.float16)
        w16 = np.ascontiguousarray(w16, dtype=np.float16)
        m, k = a16.shape
        n = w16.shape[0]
        out = np.empty((m, n), dtype=np.float32)
        self._chk(self.lib.rk_debug_gemm(self.h, a16.view(np.uint16).ctypes.data_as(_P(C.c_uint16)),
                                         w16.view(np.uint16).ctypes.data_as(_P(C.c_uint16)),
                                         out.ctypes.data_as(_f32p), m, n, k, int(use_glds)))
        return out

    def gemm_bench(self, m: int, n: int, k: int, epi: int = 0, iters: int = 20) -> float:
        """average ms per launch of the engine GEMM at (m, n, k)"""
        ms = C.c_float(0)
        self._chk(self.lib.rk_debug_gemm_bench(self.h, m, n, k, epi, iters, C.byref(ms)))
        return float(ms.value)

    def debug_read(self, name: str, n_floats: int) -> np.ndarray:
        out = np.empty(n_floats, dtype=np.float32)
        got = self.lib.rk_debug_read(self.h, name.encode(), out.ctypes.data_as(_f32p), n_floats)
        if got < 0:
            self._chk(int(got))
        return out[:got]


class RkLlamaEngine(RkEngine):
    """Decoder-only engine (rk_llama_*): prefill + last-position logits.  Shares weight loading, options, profiling and
    lifetime with RkEngine; the T5 entry points refuse it."""

    def __init__(self, dims, device: int = 0, max_tokens: int = 16384, max_seqs: int = 16):
        self.lib = load_library()
        self.dims = dims
        self.desc = RkLlamaDesc(vocab=dims.vocab, hidden=dims.hidden, n_heads=dims.n_heads, n_kv_heads=dims.n_kv_heads,
                                head_dim=dims.head_dim, intermediate=dims.intermediate, n_layers=dims.n_layers,
                                tied_head=int(dims.tied_head), eps=dims.eps, rope_theta=dims.rope_theta,
                                max_tokens=max_tokens, max_seqs=max_seqs)
        h = C.c_void_p()
        rc = self.lib.rk_llama_create(C.byref(self.desc), device, C.byref(h))
        if rc != 0:
            raise RkError(rc, (self.lib.rk_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        self.comm_rank, self.comm_world = 0, 1
        if getattr(dims, "rope_scaling", None) is not None:          # rope type llama3: before finalize builds the rotary tables
            f, lo, hi, orig = dims.rope_scaling
            self._chk(self.lib.rk_llama_set_rope_scaling(self.h, float(f), float(lo), float(hi), int(orig)))

    def greedy1(self, seqs: Sequence[Sequence[int]]) -> np.ndarray:
        tok, off = pack_ragged(seqs)
        out = np.empty(len(seqs), dtype=np.int32)
        self._chk(self.lib.rk_llama_greedy1(self.h, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs),
                                            out.ctypes.data_as(_i32p)))
        return out

    def last_logits(self, seqs: Sequence[Sequence[int]], out_ids: Sequence[int]) -> np.ndarray:
        tok, off = pack_ragged(seqs)
        oi = _i32(out_ids)
        out = np.empty((len(seqs), len(oi)), dtype=np.float32)
        self._chk(self.lib.rk_llama_last_logits(self.h, tok.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), len(seqs),
                                                oi.ctypes.data_as(_i32p), len(oi), out.ctypes.data_as(_f32p)))
        return out


def rel_bucket(rel: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> int:
    """The C++ restatement of hf: modeling_t5.py:216-262 that builds the device bias tables (host-only call)."""
    return load_library().rk_rel_bucket(int(rel), int(bidirectional), num_buckets, max_distance)
