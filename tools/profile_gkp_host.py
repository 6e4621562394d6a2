"""cProfile of the paper-setting GKP Grover run (host side): where the wall clock goes between the launches.

    python3 tools/profile_gkp_host.py [lines]
"""
import cProfile
import io
import pstats
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
lines = int(sys.argv[1]) if len(sys.argv) > 1 else 45
sys.argv = [str(ROOT / "tools" / "bench_gkp.py"), "--circuit", "grover27", "--bond", "100", "--rel-err", "1e-2"]
prof = cProfile.Profile()
prof.enable()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    prof.disable()
    for key in ("cumulative", "tottime"):
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats(key).print_stats(lines)
        print(out.getvalue())
