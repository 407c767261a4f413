import csv, glob, collections, sys, os
src=sys.argv[1]
cnt={}
for d in sorted(glob.glob(src+'/pmc_*')):
    f=glob.glob(d+'/*/*counter_collection.csv') if os.path.isdir(d) else []
    if not f: continue
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if 'fused1024' in r['Kernel_Name']:
            per[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
            meta={k:r[k] for k in ('VGPR_Count','LDS_Block_Size','Grid_Size','Workgroup_Size','Scratch_Size') if k in r}
    for c,dd in per.items():
        v=list(dd.values()); cnt[c]=sum(v)/len(v)
frames=1801600
print(meta)
for k,v in sorted(cnt.items()): print("%-28s %14.0f  per frame %.2f"%(k,v,v/frames))
wc=cnt.get('SQ_WAVE_CYCLES')
if wc:
    for k in ('SQ_WAIT_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_LDS'):
        if k in cnt: print("frac",k,"%.3f"%(cnt[k]/wc))
if 'SQ_LDS_BANK_CONFLICT' in cnt: print("lds conflict/active %.3f"%(cnt['SQ_LDS_BANK_CONFLICT']/cnt['SQ_LDS_IDX_ACTIVE']))
for kk in glob.glob(src+'/kt/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(kk)):
        if 'fused1024' in r['Name']: print("kernel trace avg ns",r['AverageNs'],"calls",r['Calls'])
