#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcF
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmcF -- scratch/ubench/valu > gpurun_out/pmcF.log 2>&1
python - <<'PY'
import csv, collections, glob
f=glob.glob('gpurun_out/pmcF/*/*_counter_collection.csv')[0]
t=glob.glob('gpurun_out/pmcF/*/*_kernel_trace.csv')[0]
dur={}
for r in csv.DictReader(open(t)):
    dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'][:40], r.get('Grid_Size_X', r.get('Grid_Size')))
agg=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    agg[r['Dispatch_Id']][r['Counter_Name']]=float(r['Counter_Value'])
for d,v in list(agg.items())[:16]:
    ns,name,grid=dur.get(d,(0,'?',0))
    w=v.get('SQ_WAVES',1)
    print(name, 'grid',grid,'ns',ns,'waves',w,'WAVE_CYCLES/wave %.0f'%(v['SQ_WAVE_CYCLES']/w),'-> per ns %.3f'%(v['SQ_WAVE_CYCLES']/w/ns), 'GUI_ACTIVE %.0f per ns %.3f'%(v['GRBM_GUI_ACTIVE'], v['GRBM_GUI_ACTIVE']/ns), 'VALU/wave %.0f active/wave %.0f'%(v['SQ_INSTS_VALU']/w, v['SQ_ACTIVE_INST_VALU']/w))
PY
