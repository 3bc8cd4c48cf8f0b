import sys, os
sys.path[:0] = ["/root/repo/llm-rankers_amd", "/root/repo"]
import torch
from llmrankers import _synth
from llmrankers._engine import RkEngine
dims = _synth.TOY_GATED_UNTIED
eng = RkEngine(dims, 0, max_tokens=2048, max_seqs=16, max_dec_len=4).load_state(_synth.synth_state_dict(dims, 3).items())
print("occupancy:", eng.debug_read("occupancy", 10).tolist())
