set -e
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/dwtrace -o r --output-format csv -- python3 tools/kernel_sequence.py x3dl > gpurun_out/kseq2.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/dwtrace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=open('gpurun_out/dwtrace_dw.txt','w')
for r in rows[-400:]:
    n=r['Kernel_Name']
    if 'dw_' in n or 'layernorm' in n or 'mlp_fused' in n:
        out.write("%s %s grid=%s wg=%s lds=%s dur_us=%.1f\n"%(r['Start_Timestamp'],n[:60],r['Grid_Size_X'],r['Workgroup_Size_X'],r['LDS_Block_Size'],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
find gpurun_out/dwtrace -name "*.csv" -delete
head -30 gpurun_out/dwtrace_dw.txt
