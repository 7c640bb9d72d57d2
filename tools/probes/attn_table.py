import json,sys
for p in sys.argv[1:]:
    d=json.loads([l for l in open(p) if l.startswith('{')][-1])
    k=d['kernels']
    print(p.split('/')[-1], 'step %.2f ms'%d['ms_per_step'], ' '.join('%s %.1f us (%.0f GB/s)'%(n,1e3*k[n]['ms_per_launch'],k[n].get('achieved',0)) for n in ('spatial','bwd_spatial','temporal','bwd_ctxgrad','bwd_reduce_T') if n in k))
