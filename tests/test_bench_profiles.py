"""bench.py's roofline block reads committed rocprofv3 summaries (profiles/<PROFILE_ROUND>_*): counter traffic, the instruction census, the
kernel table, the VALU micro-benchmark.  A file that is missing silently empties part of the driver's line (round 5: `roofline.valu.issue_frac`
vanished when PROFILE_ROUND moved on and one file did not) — so the CPU suite fails while any file a default run reads is absent."""
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_profile_file_bench_reads_is_committed():
    missing = [f for f in bench.profile_files_read() if not os.path.exists(os.path.join(ROOT, f))]
    assert not missing, "bench.py reads these and they are absent (tools/collect_all_profiles.sh %s, then copy into profiles/): %s" % (bench.PROFILE_ROUND, missing)


def test_compute_side_reading_has_its_inputs():
    """the issue-fraction estimate needs the class census, the cycle pass AND the micro-benchmark's costs: with all three present for the
    headline configuration pmc_valu() returns issue_frac"""
    assert bench._micro_costs(), bench._micro_costs_file()
    v = bench.pmc_valu("1", 50_000_000)
    assert v is not None and v.get("issue_frac") and 0.5 < v["issue_frac"] < 1.3, v
    assert v["instruction_costs_file"].startswith("profiles/")
    r = bench.rocprof_kernel_avg("1")
    assert r is not None and r["calls"] >= 9 and r["min_ms"] <= r["avg_ms_without_slowest_call"] <= r["avg_ms"], r
