#!/bin/bash
# The profiles of one round, in one gpurun call on one box:
#     gpurun --timeout 3000 -- 'bash tools/profile_round.sh r04'
# writes gpurun_out/<tag>/..., condenses into profiles/<tag>_* (tools/profile_summary.py) -- copy what gpurun merges back
# under gpurun_out/<tag>/profiles/ into profiles/ afterwards (the box's own profiles/ does not travel back).
# Needs the instrumented library for the timelines (tools/timeline.py builds it with -DLENTIL_TIMELINE when it is missing).
set -u
cd "$(dirname "$0")/.."
TAG=${1:-rXX}
O=gpurun_out/$TAG; mkdir -p $O/profiles
export TMPDIR=/tmp
# (the runtime reads it when the profiler's preloaded library initialises it -- before bench.py's own os.environ.setdefault runs)
export GPU_MAX_HW_QUEUES=8
B="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone"
B4="$B --lens petzval_58mm --aovs 8"                       # BASELINE config 4
# 1. per-kernel durations of the headline command and of config 4 (the program straight after --)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_stats   -- python3 $B  > $O/stats_headline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}c4_stats -- python3 $B4 > $O/stats_config4.log 2>&1
# 2. HBM traffic of the scan kernels, separate --pmc passes.  PMC passes serialise kernels, so the chunked form with one
#    whole-frame scan launch (the streamed pass's resident solve waves would wait 250 ms for a scan that is not running)
export LENTIL_STREAM=0 LENTIL_CHUNKS=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/${TAG}_fetch   -- python3 $B  > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/${TAG}_write   -- python3 $B  > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/${TAG}c4_fetch -- python3 $B4 > $O/pmc_fetch_c4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/${TAG}c4_write -- python3 $B4 > $O/pmc_write_c4.log 2>&1
unset LENTIL_STREAM LENTIL_CHUNKS
# (the tags bench.py looks for in profiles/pmc_scan_latest.json: its workload_tag)
W1="scan_dma2_kernel double_gauss_50mm 3840x2160 M=9 samples=1024 aovs=1 f_hi=1.53e-05"
W4="scan_dma_multi_kernel petzval_58mm 3840x2160 M=9 samples=1024 aovs=9 f_hi=1.53e-05"
python3 tools/profile_summary.py ${TAG}c4 /tmp/${TAG}c4_stats /tmp/${TAG}c4_fetch /tmp/${TAG}c4_write --workload "$W4" > $O/summary_config4.txt 2>&1
# (the headline last: profiles/pmc_scan_latest.json, which step 6's bench line reads for roofline.traffic, is the last one written)
python3 tools/profile_summary.py $TAG   /tmp/${TAG}_stats   /tmp/${TAG}_fetch   /tmp/${TAG}_write   --workload "$W1" > $O/summary_headline.txt 2>&1
find /tmp/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $O/profiles/${TAG}_rocprofv3_kernel_stats_full.csv \;
# the scan kernel's launches one by one (first two of a context: the half-frame launches of its first, chunked pass)
python3 - "$TAG" > $O/profiles/${TAG}_scan_launches.txt <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for d, name in (("/tmp/%s_stats" % tag, "headline"), ("/tmp/%sc4_stats" % tag, "config 4")):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if "scan_" in r["Kernel_Name"]]
    print(name, ":", rows[0]["Kernel_Name"].split("(")[0] if rows else "-")
    print("  launch durations, us:", " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows))
PY
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}c4_kernel_stats.csv profiles/${TAG}_pmc_scan.json profiles/${TAG}c4_pmc_scan.json profiles/pmc_scan_latest.json $O/profiles/ 2>/dev/null
# 3. timelines of one streamed pass (instrumented build)
python3 tools/timeline.py --passes 7 --out $O/profiles/${TAG}_timeline_headline.txt > $O/timeline_headline.log 2>&1
python3 tools/timeline.py --lens petzval_58mm --aovs 8 --passes 5 --out $O/profiles/${TAG}_timeline_config4.txt > $O/timeline_config4.log 2>&1
python3 tools/timeline.py --width 7680 --height 4320 --samples 2048 --passes 4 --out $O/profiles/${TAG}_timeline_config5.txt > $O/timeline_config5.log 2>&1
# 4. cryptomatte replay
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG}_cr -- python3 tools/crypto_rate.py > $O/crypto_stats.log 2>&1
python3 - /tmp/${TAG}_cr $O/profiles/${TAG}_crypto_kernels.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
with open(sys.argv[2], "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats of tools/crypto_rate.py: the pass without cryptomatte AOVs, with a draw log only, with one and with three\n")
    for r in (csv.DictReader(open(f[0])) if f else []):
        if "crypto_" in r["Name"] or "flag_bits" in r["Name"]:
            out.write("%-40s calls %4s  avg %10.1f us  min %10.1f  max %10.1f\n" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
GPU_MAX_HW_QUEUES=8 python3 tools/crypto_rate.py > $O/profiles/${TAG}_crypto_rate.json 2> $O/crypto_rate.err
# 5. config 5 on one GPU, streamed and chunked
C5="bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-regime --no-configs --no-pcie --no-parity-check --no-scan-alone --width 7680 --height 4320 --samples 2048"
python3 $C5 2>/dev/null | tail -1 > $O/profiles/${TAG}_config5_one_gpu_streamed.json
LENTIL_STREAM=0 python3 $C5 2>/dev/null | tail -1 > $O/profiles/${TAG}_config5_one_gpu_chunked.json
# 6. single bands of the multi-GPU frames, each alone on this GPU
bash tools/emulate_bands.sh > $O/profiles/${TAG}_emulated_bands.txt 2>&1
# 7. the bench line as the driver runs it
python3 bench.py --steps 20 --warmup 3 > $O/profiles/${TAG}_bench_line.json 2> $O/bench.err
ls -la $O/profiles
