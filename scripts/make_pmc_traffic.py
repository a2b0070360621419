"""Build profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, as the
MI355X guide prescribes) of the bench command (256 frames per launch).

On the GPU box:
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline
then (anywhere):  python scripts/make_pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json
"""
import csv, glob, json, os, sys
from collections import defaultdict

STAGES = {  # kernel-name substring -> bench.py stage name
    "frame_min_kernel": "frame_min",
    "blur_mfma_kernel<61": "blur_l0_k61x61", "blur_band_kernel<61": "blur_l0_k61x61",
    "blur_mfma_kernel<33": "blur_l1_k33x33", "blur_band_kernel<33": "blur_l1_k33x33",
    "blur_mfma_kernel<17": "blur_l2_k17x17", "blur_band_kernel<17": "blur_l2_k17x17",
    "taxim_tail_kernel": "tail_fused",
}
FRAMES, NPIX = 256, 320 * 240


def collect(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            for sub, stage in STAGES.items():
                if sub in row["Kernel_Name"]:
                    acc[stage] += float(row["Counter_Value"]); cnt[stage] += 1
                    break
    return {k: acc[k] / cnt[k] for k in acc}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {
    "_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on `python bench.py --steps 5 --warmup 2 "
                   "--no-cpu-baseline --no-roofline` (256 envs through GelSightSensor.update()), MI355X; built by scripts/make_pmc_traffic.py. Counter unit KiB. FETCH_SIZE is doubled "
                   "(gfx950 reports exactly 1/2 of wide coalesced streaming reads, MI355X_MICROARCH.md HBM section; check: frame_min reads "
                   "78.6 MB, counter 39.3 MB). Values are HBM bytes PER FRAME (320x240); bench.py multiplies by the frames per launch.",
    "frames_per_launch_measured": FRAMES,
    "per_frame_bytes": {k: int((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 / FRAMES) for k in sorted(set(fetch) | set(write))},
    "raw_kib_per_launch": {k: {"FETCH_SIZE": int(fetch.get(k, 0)), "WRITE_SIZE": int(write.get(k, 0))} for k in sorted(set(fetch) | set(write))},
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["per_frame_bytes"], indent=1))
