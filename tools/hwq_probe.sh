B="python bench.py --no-cpu-baseline --no-parity --regions 5 --steps 200"
show() { python -c "import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('$2', round(d['value']/1e3,1),'k six-stream', round(d['single_stream']['value']/1e3,1),'k single')"; }
for q in default 3 4 5; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  $B > $OUT/plain_$q.json 2>/dev/null; show $OUT/plain_$q.json "plain       GPU_MAX_HW_QUEUES=$q"
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 $B > $OUT/dist_$q.json 2>/dev/null; show $OUT/dist_$q.json "distributed GPU_MAX_HW_QUEUES=$q"
done
unset GPU_MAX_HW_QUEUES
for s in 3 9; do $B --streams $s > $OUT/plain_s$s.json 2>/dev/null; show $OUT/plain_s$s.json "plain default queues, --streams $s"; done
