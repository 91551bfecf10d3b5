"""Copy the judged summaries out of gpurun_out/ (scratch) into profiles/ (tracked).
usage: python tools/collect_profiles.py <tag> <bench_json> <kernel_stats_dir> <pmc_fetch_dir> <pmc_write_dir>"""
import collections, csv, glob, json, shutil, sys
tag, bench, stats_dir, fdir, wdir = sys.argv[1:6]
shutil.copy(bench, f"profiles/{tag}_bench.json")
shutil.copy((glob.glob(f"{stats_dir}/*kernel_stats.csv") + glob.glob(f"{stats_dir}/*/*kernel_stats.csv"))[0], f"profiles/{tag}_msm2p22_kernel_stats.csv")
out = {}
lines = ["kernel,counter,launches,avg_value_KB"]
for c, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open((glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0])):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        lines.append(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.1f}")
        out[(k, c)] = sum(v) / len(v)
open(f"profiles/{tag}_msm2p22_pmc_fetch_write.csv", "w").write("\n".join(lines) + "\n")
f = [v for (k, c), v in out.items() if "k_accumulate" in k and c == "FETCH_SIZE"][0]
w = [v for (k, c), v in out.items() if "k_accumulate" in k and c == "WRITE_SIZE"][0]
dg = [v for (k, c), v in out.items() if "k_digits" in k and c == "FETCH_SIZE"][0]
entries_bytes = 16 * (1 << 22) * 4           # the sorted-entry stream of the 2^22 x 16-window launch
traffic = int(f * 1024 + entries_bytes / 2 + w * 1024)
json.dump({"k_accumulate_2p22": traffic,
           "_how": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{tag}_msm2p22_pmc_fetch_write.csv). "
                   "FETCH_SIZE on gfx950 tallies requests at 64 B each: a streaming kernel that reads full 128-B lines is reported at half its bytes "
                   f"(this run's k_digits: 128 MiB of scalars read, {dg / 1024:.1f} MiB reported), a 64-byte gather at its true bytes "
                   "(tools/gather_probe.hip, profiles/r01_h_gather_calibration.txt: 768 MiB of random 64-B records moved, 764 MiB reported). "
                   "k_accumulate is 64-byte base gathers plus a 4-byte entry stream, so bytes = FETCH_SIZE * 1024 + half of the entry stream (the part the "
                   "counter misses) + WRITE_SIZE * 1024",
           "_raw_KB": {"FETCH_SIZE": f, "WRITE_SIZE": w}}, open("profiles/pmc_traffic.json", "w"), indent=1)
print("k_accumulate traffic GB:", traffic / 1e9)
