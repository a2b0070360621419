"""Build profiles/pmc_traffic_rNN.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, as the
MI355X guide prescribes) of the bench command.

On the GPU box (scripts/gpu_round.sh <tag> pmc does exactly this):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-roofline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-roofline
then (anywhere):  python scripts/make_pmc_traffic.py <pmc_fetch dir> <pmc_write dir> <out.json> [frames_per_tail_launch=0: from the tail's WRITE_SIZE] [H=240] [W=320] [commit] [frames_per_shard]
"""
import csv, glob, json, os, sys
from collections import defaultdict

STAGES = {  # kernel-name substring -> bench.py stage name
    "frame_min_kernel<true>": "frame_min_from_depth", "frame_min_kernel<false>": "frame_min",
    "blur_mfma_kernel<61": "blur_l0_k61x61", "blur_band_kernel<61": "blur_l0_k61x61",
    "blur_mfma_kernel<33": "blur_l1_k33x33", "blur_band_kernel<33": "blur_l1_k33x33",
    "blur_mfma_kernel<17": "blur_l2_k17x17", "blur_band_kernel<17": "blur_l2_k17x17",
    "blur_band_loop_kernel": "blur_l0_k117x117", "blur_mfma_kernel<117": "blur_l0_k117x117", "blur_mfma_kernel<15": "blur_l3_k15x15",
    "frame_rows_kernel<true>": "frame_min_from_depth", "frame_rows_kernel<false>": "frame_min",
    "taxim_stream_kernel": "tail_fused", "taxim_tail_kernel": "tail_fused_tiled",
}
FRAMES = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 0 = derive from the tail's own WRITE_SIZE (12 B/px of RGB per frame)
H = int(sys.argv[5]) if len(sys.argv) > 5 else 240
W = int(sys.argv[6]) if len(sys.argv) > 6 else 320
COMMIT = sys.argv[7] if len(sys.argv) > 7 else "unknown"
BATCH = int(sys.argv[8]) if len(sys.argv) > 8 else FRAMES  # frames per depth -> height-map dispatch (the sensor's whole shard)


def frames_per_tail_launch(write_dir):
    """Frames one tail launch renders, from what it WROTE: RGB is 12 B/px and nothing else of size leaves the kernel, so WRITE_SIZE
    per dispatch / (12 H W) is the frame count whatever pass policy the library chose.  (Round 3's 640x480 file was built with the
    320x240 default of 1024 frames where the 640x480 pass launches 256: every per-frame figure in it was 4x too small.)"""
    vals = {"taxim_stream_kernel": [], "taxim_tail_kernel": []}
    for f in glob.glob(os.path.join(write_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == "WRITE_SIZE":
                for k in vals:
                    if k in row["Kernel_Name"]:
                        vals[k].append(float(row["Counter_Value"]))
    v = vals["taxim_stream_kernel"] or vals["taxim_tail_kernel"]  # (the tiled tail only runs the sensors' reset renders: not the pass size)
    if not v:
        raise SystemExit("no tail dispatch in the WRITE_SIZE pass")
    v.sort()
    return v[len(v) // 2] * 1024 / (12.0 * H * W)  # median dispatch


_fpl = frames_per_tail_launch(sys.argv[2])
if FRAMES == 0:
    # the tail also writes a few KB of observation partial sums and FOTS records per frame (+2-3 %): passes are multiples of 64 frames
    FRAMES = max(64, int(round(_fpl / 64.0)) * 64)
    if abs(_fpl / FRAMES - 1.0) > 0.06:
        raise SystemExit(f"cannot tell the frames per tail launch from its writes ({_fpl:.1f} frames of {W}x{H} RGB per dispatch)")
elif abs(_fpl / FRAMES - 1.0) > 0.1:
    raise SystemExit(f"frames_per_tail_launch={FRAMES} contradicts the tail's own writes ({_fpl:.1f} frames of {W}x{H} RGB per dispatch)")
if len(sys.argv) <= 8:
    BATCH = FRAMES


def depth_frames_total(write_dir):
    """Frames the depth -> height map dispatches of the WRITE_SIZE pass converted, from what they wrote (4 B/px of height map per frame;
    the bench does not ask for the uint8 camera image): the pass may run as one launch per shard or as one launch per band-level chunk."""
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(write_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == "WRITE_SIZE" and ("frame_rows_kernel<true>" in row["Kernel_Name"] or "frame_min_kernel<true>" in row["Kernel_Name"]):
                tot += float(row["Counter_Value"]); n += 1
    return tot * 1024 / (4.0 * H * W) if n else 0.0


DEPTH_FRAMES_TOTAL = depth_frames_total(sys.argv[2])


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
    # Per-FRAME counter value.  The band levels may be launched over sub-ranges of a pass (Infinity-Cache-sized pieces), so their
    # totals are divided by the frames the run rendered = tail launches x frames per tail launch; every other kernel processes
    # FRAMES frames per dispatch.
    n_pass = cnt.get("tail_fused", 0) + cnt.get("tail_fused_tiled", 0)  # every pass of FRAMES frames ends in exactly one tail launch
    out = {}
    for k in acc:
        if n_pass > 0 and k.startswith("blur_"):
            out[k] = acc[k] / (n_pass * FRAMES)
        elif k == "frame_min_from_depth" and DEPTH_FRAMES_TOTAL:
            # the depth pass deferred into the render (ABI 11) runs per band-level chunk: total over the frames it converted
            out[k] = acc[k] / DEPTH_FRAMES_TOTAL
        else:
            out[k] = acc[k] / cnt[k] / (BATCH if k.startswith("frame_min") else FRAMES)
    return out, dict(cnt)


(fetch, fcnt), (write, wcnt) = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
keys = sorted(set(fetch) | set(write))
per_frame = {k: int((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024) for k in keys}
path = [k for k in keys if k.startswith("blur_") or k == "tail_fused"]
out = {
    "_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on `python3 bench.py --steps 3 --warmup 2 "
                   "--no-cpu-baseline --no-sweep --no-roofline` (the driver's headline workload through GelSightSensor.update()), MI355X; built by "
                   "scripts/make_pmc_traffic.py.  Counter unit KiB.  FETCH_SIZE is doubled (gfx950 reports exactly 1/2 of wide coalesced "
                   "streaming reads, MI355X_MICROARCH.md HBM section; check: the depth -> height-map pass reads 4 B/px and its doubled "
                   "counter says so).  Values are HBM bytes PER FRAME; bench.py multiplies by the frames per launch.",
    "measured_at_commit": COMMIT,
    "frames_per_tail_launch_measured": FRAMES, "resolution": [W, H],
    "per_frame_bytes": per_frame,
    "taxim_path_sum_per_frame": sum(per_frame[k] for k in path),
    "compulsory_per_frame_16B_per_px": 16 * H * W,
    "raw_kib_per_frame": {k: {"FETCH_SIZE": round(fetch.get(k, 0), 2), "WRITE_SIZE": round(write.get(k, 0), 2)} for k in keys},
    "dispatches": {k: {"fetch_pass": fcnt.get(k, 0), "write_pass": wcnt.get(k, 0)} for k in keys},
}
out["frames_per_tail_launch_from_writes"] = round(_fpl, 2)
if out["taxim_path_sum_per_frame"] < out["compulsory_per_frame_16B_per_px"]:
    raise SystemExit(f"path sum {out['taxim_path_sum_per_frame']} B/frame is BELOW the compulsory {out['compulsory_per_frame_16B_per_px']}: wrong normalisation")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("per_frame_bytes", "taxim_path_sum_per_frame", "compulsory_per_frame_16B_per_px")}, indent=1))
