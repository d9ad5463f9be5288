ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/qrtrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TN_QR_NBO=${1:-256}
rocprofv3 --kernel-trace --output-format csv -d $OUT -o qr -- python3 $ROOT/tools/qr_trace_one.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/*kernel_trace.csv')+glob.glob('$OUT/*/*kernel_trace.csv')
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# second QR only: last half
sel=[r for r in rows if 'tn::' in r['Kernel_Name']]
half=sel[len(sel)//2:]
out=open('$OUT/second_qr_dispatches.txt','w')
for r in half:
    name=r['Kernel_Name'].split('(')[0][-48:]
    out.write('%-50s %9.1f us  grid %s wg %s\n'%(name,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Grid_Size_X','?'),r.get('Workgroup_Size_X','?')))
out.close()
big=[ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in half if 'gemm_kernel<128, 128' in r['Kernel_Name']]
print('gemm128 launches',len(big),'total %.2f ms'%(sum(big)/1e3),'top',sorted(big)[-12:])
PY
rm -f $OUT/*kernel_trace.csv $OUT/*/*kernel_trace.csv
