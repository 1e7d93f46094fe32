"""bench.py's per-stage table and the probes under tools/ name C entry points as strings: every such name must be a symbol the
library exports (svgp-vae_amd/_lib.py SIGNATURES = include/svgpvae_hip.h), so that a renamed or removed entry point fails HERE and
not on the GPU box at the end of a round."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _names(path):
    src = open(os.path.join(ROOT, path)).read()
    return set(re.findall(r'"(svgp_[a-z0-9_]+)"', src))


def test_every_entry_point_named_by_bench_and_the_probes_is_exported():
    from svgp_vae_amd import _lib
    known = set(_lib.SIGNATURES)
    for path in ("bench.py", "tools/decoder_split_probe.py"):
        names = _names(path)
        assert names, path
        missing = sorted(n for n in names if n not in known)
        assert not missing, (path, missing)


def test_stage_table_covers_the_schedule_of_the_single_gpu_step():
    """The m <= 64 table must name the merged launches the step really issues (csrc/api.hip), not their unmerged predecessors."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    api = open(os.path.join(ROOT, "svgp-vae_amd", "csrc", "api.hip")).read()
    for sym in ("svgp_gp_stats_factor_bwd_wgrad", "svgp_mnist_encoder_bwd_km_sum", "svgp_gp_posterior_bwd_rows",
                "svgp_mnist_decoder_bwd_data_pre_aji", "svgp_mnist_decoder_fwd_pre"):
        assert sym in src, sym
        assert sym in api, sym
