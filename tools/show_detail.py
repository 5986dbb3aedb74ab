import json,sys
d=json.load(open(sys.argv[1]))
print('ms_per_step', d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['config'].get('hip_launches_per_step'))
hk=d.get('hip_kernels',{})
rows=sorted(((v.get('ms_per_step',0),n,v) for n,v in hk.items()), reverse=True)
for ms,n,v in rows[:18]: print('%-30s launches %5.0f  ms/step %6.2f  avg us %7.1f' % (n, v['launches_per_step'], ms, v['avg_launch_us']))
