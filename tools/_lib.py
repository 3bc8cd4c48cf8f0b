"""tools/ only: run a tool against a separately compiled engine library (A/B builds with other -D switches, -DRK_MEASURE
measurement builds with timing knock-outs).  `RK_ENGINE_LIB=<path> python tools/<tool>.py ...` - the product package itself
ignores that variable (llmrankers/_engine.py always loads the in-tree library)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_env_library():
    """Make $RK_ENGINE_LIB (if set) the library every RkEngine of this process binds.  Returns the path or None."""
    path = os.environ.get("RK_ENGINE_LIB")
    if not path:
        return None
    sys.path[:0] = [os.path.join(REPO, "llm-rankers_amd")]
    from llmrankers import _engine
    _engine.load_library(os.path.abspath(path), make_default=True)
    print(f"[tools] engine library: {path}", file=sys.stderr)
    return path
