"""Copy the judged summaries of one tools/gpu_profile_all.sh run out of gpurun_out/ (scratch) into profiles/
(tracked) and regenerate profiles/pmc_traffic.json from its FETCH_SIZE / WRITE_SIZE passes.
usage: python tools/collect_profiles.py <tag> [<file tag, default = tag>]      (reads gpurun_out/*_<tag>*)"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
ftag = sys.argv[2] if len(sys.argv) > 2 else tag
G = "gpurun_out"


def first(pattern):
    hits = glob.glob(pattern) + glob.glob(pattern.replace("/*", "/*/*"))
    return hits[0] if hits else None


def counters(d, counter):
    """-> {kernel name without arguments: [value per dispatch, KB]}"""
    agg = collections.defaultdict(list)
    f = first(f"{d}/*counter_collection.csv")
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


def copy(src, dst):
    if src and os.path.exists(src):
        shutil.copy(src, dst)
        print("  ", dst)


copy(f"{G}/prof_{tag}.json", f"profiles/{ftag}_bench.json")
copy(first(f"{G}/prof_{tag}/*kernel_stats.csv"), f"profiles/{ftag}_msm2p22_kernel_stats.csv")
copy(first(f"{G}/profntt_{tag}/*kernel_stats.csv"), f"profiles/{ftag}_ntt2p24_kernel_stats.csv")
copy(f"{G}/pmcmsm_{tag}.txt", f"profiles/{ftag}_msm2p22_pmc_valu.txt")
copy(f"{G}/pmcntt_{tag}.txt", f"profiles/{ftag}_ntt2p24_pmc.txt")
copy(f"{G}/bench_full_{tag}.json", f"profiles/{ftag}_bench_full.json")

traffic = {}
# ---- MSM: k_accumulate of the 2^22 headline ------------------------------------------------------------------
lines = ["kernel,counter,launches,avg_value_KB"]
out = {}
for c, d in (("FETCH_SIZE", f"{G}/pmcf_{tag}"), ("WRITE_SIZE", f"{G}/pmcw_{tag}")):
    for k, v in counters(d, c).items():
        lines.append(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.1f}")
        out[(k, c)] = sum(v) / len(v)
if out:
    open(f"profiles/{ftag}_msm2p22_pmc_fetch_write.csv", "w").write("\n".join(lines) + "\n")
    f = [v for (k, c), v in out.items() if "k_accumulate" in k and c == "FETCH_SIZE"][0]
    w = [v for (k, c), v in out.items() if "k_accumulate" in k and c == "WRITE_SIZE"][0]
    dg = [v for (k, c), v in out.items() if "k_digits" in k and c == "FETCH_SIZE"][0]
    entries_bytes = 16 * (1 << 22) * 4           # the sorted-entry stream of the 2^22 x 16-window launch
    traffic["k_accumulate_2p22"] = int(f * 1024 + entries_bytes / 2 + w * 1024)
    traffic["_how"] = (f"rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{ftag}_msm2p22_pmc_fetch_write.csv). "
                       "FETCH_SIZE on gfx950 tallies requests at 64 B each: a streaming kernel that reads full 128-B lines is reported at half its bytes "
                       f"(this run's k_digits: 128 MiB of scalars read, {dg / 1024:.1f} MiB reported), a 64-byte gather at its true bytes "
                       "(tools/gather_probe.hip, profiles/r01_h_gather_calibration.txt: 768 MiB of random 64-B records moved, 764 MiB reported). "
                       "k_accumulate is 64-byte base gathers plus a 4-byte entry stream, so bytes = FETCH_SIZE * 1024 + half of the entry stream (the part the "
                       "counter misses) + WRITE_SIZE * 1024")
    traffic["_raw_KB"] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    print("k_accumulate traffic GB:", traffic["k_accumulate_2p22"] / 1e9)
# ---- NTT 2^24: the three k_ntt_wave passes of one transform ---------------------------------------------------
lines = ["kernel,counter,launches,avg_value_KB,min_KB,max_KB"]
ntt = {}
for c, d in (("FETCH_SIZE", f"{G}/pmcfntt_{tag}"), ("WRITE_SIZE", f"{G}/pmcwntt_{tag}")):
    for k, v in counters(d, c).items():
        lines.append(f"\"{k}\",{c},{len(v)},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f}")
        if "k_ntt_wave" in k:
            ntt[(c, "pass1" if ", 1>" in k else "pass23")] = sum(v) / len(v)
if ntt:
    open(f"profiles/{ftag}_ntt2p24_pmc_fetch_write.csv", "w").write("\n".join(lines) + "\n")
    fetch_raw = ntt[("FETCH_SIZE", "pass1")] + 2 * ntt[("FETCH_SIZE", "pass23")]
    write_raw = ntt[("WRITE_SIZE", "pass1")] + 2 * ntt[("WRITE_SIZE", "pass23")]
    traffic["ntt_2p24"] = int(2 * fetch_raw * 1024 + write_raw * 1024)
    traffic["_how_ntt"] = (f"per transform = pass 1 (k_ntt_wave<Fr29, 1>) + passes 2 and 3 (<Fr29, 3>, average of the two), profiles/{ftag}_ntt2p24_pmc_fetch_write.csv. "
                           "Every read of the NTT is a coalesced stream (tile rows of 128 contiguous bytes, the 48-byte table entries of consecutive elements), which "
                           "FETCH_SIZE reports at half its bytes: pass 1 reads 512 MiB of data + 768 MiB of first-post-twiddle table = 1280 MiB and is reported at "
                           f"{ntt[('FETCH_SIZE', 'pass1')] / 1024:.0f} MiB; so bytes = 2 x FETCH_SIZE + WRITE_SIZE (16-byte stores: exact)")
    traffic["_raw_KB_ntt"] = {"FETCH_SIZE": fetch_raw, "WRITE_SIZE": write_raw}
    print("ntt 2^24 traffic GB:", traffic["ntt_2p24"] / 1e9)
if traffic:
    json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
