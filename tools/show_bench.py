import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d['roofline']; de=r['stage']['dense_after_timed_region']
    print(f, 'value',round(d['value']), 'mean_launch',round(r['mean_launch_us'],2),'frac',round(r['frac'],3),'behind',round(de['one_launch_pass_us_events_read_behind_each_pass'],2),'knn_alone',r.get('knn_kernel_alone_us'),'kus',r.get('kernel_us_per_step'))
