import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'value %.3fM'%(d['value']/1e6), {k:round(v['ms'],2) for k,v in d.items() if k.startswith('train_step')}, (d['roofline'].get('traffic_note') or '')[:8])
