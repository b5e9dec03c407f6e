import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], {k:(d[k].get("rays_per_s"), d[k].get("mlp_kernels_share_of_step")) for k in ("train_8192","train","train_fern","train_graphed") if k in d})
    except Exception as e: print(f,"ERR",e)
