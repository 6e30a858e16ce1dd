import csv, collections, glob, sys
def agg(d):
    out=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(int); dur=collections.defaultdict(float)
    path=glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    seen=set()
    for row in csv.DictReader(open(path)):
        k=row['Kernel_Name'].split('(')[0].replace('void ','')
        out[k][row['Counter_Name']]+=float(row['Counter_Value'])
        if row['Dispatch_Id'] not in seen:
            seen.add(row['Dispatch_Id']); cnt[k]+=1; dur[k]+=float(row['End_Timestamp'])-float(row['Start_Timestamp'])
    return out,cnt,dur
cells=2048*2048*50
rd,cnt,dur=agg('pmc_rd'); hit,_,_=agg('pmc_hit'); sq,_,_=agg('pmc_sq')
keys=[k for k in rd if k.startswith('k')]
print(f"{'kernel':22s} {'n':>4s} {'ms/call':>8s} {'rdB/cell':>9s} {'%128B':>6s} {'L2hit%':>7s} {'vmemrd/thr':>10s} {'valu/thr':>9s} {'waitInst%':>9s} {'waitAny%':>9s} {'active%':>8s}")
for k in sorted(keys,key=lambda k:-dur[k]):
    n=cnt[k]; r=rd[k]
    b=(128*r['TCC_EA0_RDREQ_128B_sum']+64*r['TCC_EA0_RDREQ_64B_sum']+32*r['TCC_EA0_RDREQ_32B_sum'])/n
    p128=128*r['TCC_EA0_RDREQ_128B_sum']/n/max(b,1)*100
    h=hit[k]; hr=h['TCC_HIT_sum']/max(h['TCC_HIT_sum']+h['TCC_MISS_sum'],1)*100
    s=sq[k]; wv=max(s['SQ_WAVES'],1); wc=max(s['SQ_WAVE_CYCLES'],1)
    print(f"{k[:22]:22s} {n:4d} {dur[k]/n/1e6:8.3f} {b/cells:9.1f} {p128:6.1f} {hr:7.1f} {s['SQ_INSTS_VMEM_RD']/wv:10.1f} {s['SQ_INSTS_VALU']/wv:9.1f} {s['SQ_WAIT_INST_ANY']/wc*100:9.1f} {s['SQ_WAIT_ANY']/wc*100:9.1f} {s['SQ_ACTIVE_INST_ANY']/wc*100:8.1f}")
