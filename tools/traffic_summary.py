"""HBM-side traffic of the conv kernel family per training step from two rocprofv3 PMC passes.
usage: python tools/traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [steps executed] > profiles/rNN_traffic.json
Steps default to the number of nchw_to_nhwc_kernel launches in the fetch pass (one per executed forward pass: warm-up + timed steps +
the host-cost step + the two instrumented passes of bench.py); a hand-passed count went stale when bench.py gained a step.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of wide coalesced reads and is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import json
import sys

FAMILY = ('conv_igemm', 'conv_wgrad', 'conv_x3', 'splitk_epilogue', 'slab_reduce')    # splitk_epilogue: round 1 only


def total(path, counter):
    fam = allk = 0.0
    launches = passes = 0
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row['Counter_Name'] != counter:
                continue
            v = float(row['Counter_Value']) * 1024.0
            allk += v
            if 'nchw_to_nhwc_kernel' in row['Kernel_Name']:
                passes += 1
            if any(f in row['Kernel_Name'] for f in FAMILY):
                fam += v
                launches += 1
    return fam, allk, launches, passes


fetch, write = sys.argv[1], sys.argv[2]
rf, ra, n, passes = total(fetch, 'FETCH_SIZE')
wf, wa, _, wpasses = total(write, 'WRITE_SIZE')
steps = float(sys.argv[3]) if len(sys.argv) > 3 else float(passes)
assert steps > 0 and (len(sys.argv) > 3 or passes == wpasses), (passes, wpasses)
rf, ra = 2 * rf, 2 * ra
print(json.dumps({
    'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 1 --no-graph --no-inference '
              '--no-cpu-baseline`, %g steps executed; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads), '
              'units KiB -> bytes' % steps,
    'conv_family_hbm_bytes_per_step': (rf + wf) / steps,
    'conv_family_read_bytes_per_step': rf / steps,
    'conv_family_write_bytes_per_step': wf / steps,
    'conv_family_launches_per_step': n / steps,
    'all_kernels_hbm_bytes_per_step': (ra + wa) / steps,
}, indent=1))
