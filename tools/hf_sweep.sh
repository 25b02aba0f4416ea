# host-fed sweep (development aid).  HF_CFGS="streams:lanes:groups[:hwq] ..." HF_KIND=c4|c5 HF_PROF=1 (rocprofv3 trace of the first config)
mkdir -p gpurun_out
: > gpurun_out/hf_sweep.txt
for cfg in ${HF_CFGS:-256:3:40 256:2:40 256:4:40 128:4:80 512:3:24}; do
 IFS=: read st la gr hwq <<< "$cfg"
 ( [ -n "$hwq" ] && export GPU_MAX_HW_QUEUES=$hwq
 timeout 300 python bench.py --host-fed-only ${HF_KIND:-c4} --feed-streams $st --feed-lanes $la --feed-groups $gr 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hwq ${hwq:-default} streams',d['streams_per_group'],'lanes',d['lanes'],'groups',d['groups'],'M/s %.2f'%(d['value']/1e6),'sec %.3f'%d['seconds'],'up %.1f GB/s (copying %.1f)'%(d['pcie_GBps']['up_sustained'],d['pcie_GBps']['up_while_copying']),'dev ms/group %.2f up %.2f'%(d['device_ms_per_group'],d['upload_ms_per_group']), d['parity_sample']['mismatches'])" ) >> gpurun_out/hf_sweep.txt 2>&1
done
cat gpurun_out/hf_sweep.txt
if [ -n "$HF_PROF" ]; then
 set -- ${HF_CFGS:-256:3:40}; IFS=: read st la gr hwq <<< "$1"
 [ -n "$hwq" ] && export GPU_MAX_HW_QUEUES=$hwq
 cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/hf_prof -o hf -- python $GRAFT_REPO_ROOT/bench.py --host-fed-only ${HF_KIND:-c4} --feed-streams $st --feed-lanes $la --feed-groups 20 > /dev/null 2>&1
fi
