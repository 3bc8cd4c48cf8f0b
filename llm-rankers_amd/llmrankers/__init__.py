"""llmrankers — MI355X-native drop-in for the hot path of ielab/llm-rankers.

Same import names as the reference package (`llmrankers.rankers`, `llmrankers.pointwise`, `llmrankers.setwise`)
so existing scripts keep working; underneath, the T5 forward runs in hand-written HIP kernels for gfx950
through the C ABI in include/rk_engine.h.  Scope and what is deliberately absent: DESIGN.md.
"""
__version__ = "0.1.0"
