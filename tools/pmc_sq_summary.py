#!/usr/bin/env python3
"""Per-kernel issue/latency picture from the 'sq' and 'tcp' passes of tools/pmc_pass.sh (gpurun_out/pmc_sq, gpurun_out/pmc_tcp)."""
import csv, collections, glob

def agg(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
    path = glob.glob(f"gpurun_out/pmc_{d}/*counter_collection.csv")[0]
    seen = set()
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        out[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"]); cnt[k] += 1; dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return out, cnt, dur

sq, cnt, dur = agg("sq"); tc, cnt2, dur2 = agg("tcp")
print(f"{'kernel':32s} {'n':>5s} {'ms':>7s} {'waves':>8s} {'cyc/wave':>9s} {'vmemrd/w':>9s} {'lds/w':>7s} {'ldsconf%':>8s} {'waitAny%':>8s} {'waitInst%':>9s} {'active%':>8s} {'L1->L2 rd GB':>12s} {'L1hit%':>7s} {'L2hit%':>7s}")
for k in sorted(sq, key=lambda k: -dur[k]):
    if not k.startswith("k"): continue
    s = sq[k]; n = cnt[k]; wv = max(s["SQ_WAVES"], 1); wc = max(s["SQ_WAVE_CYCLES"], 1)
    t = tc.get(k, {}); n2 = max(cnt2.get(k, 1), 1)
    rd_req = t.get("TCP_TCC_READ_REQ_sum", 0.0); acc = t.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
    hit = t.get("TCC_HIT_sum", 0.0); miss = t.get("TCC_MISS_sum", 0.0)
    print(f"{k[:32]:32s} {n:5d} {dur[k]/n/1e6:7.3f} {wv/n:8.0f} {wc/wv:9.0f} {s['SQ_INSTS_VMEM_RD']/wv:9.1f} {s['SQ_INSTS_LDS']/wv:7.1f} "
          f"{s['SQ_LDS_BANK_CONFLICT']/wc*100:8.1f} {s['SQ_WAIT_ANY']/wc*100:8.1f} {s['SQ_WAIT_INST_ANY']/wc*100:9.1f} {s['SQ_ACTIVE_INST_ANY']/wc*100:8.1f} "
          f"{rd_req*64/n2/1e9:12.3f} {(1-rd_req/max(acc,1))*100:7.1f} {hit/max(hit+miss,1)*100:7.1f}")
