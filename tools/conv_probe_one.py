"""One layer of tools/conv_probe.py for counter passes: conv_probe_one.py <layer index> [f32] [reps]."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = [sys.argv[0], "500"] + sys.argv[2:3] + [sys.argv[1]]
os.environ["CONV_PROBE_ONLY"] = sys.argv[-1]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_probe.py")).read())
