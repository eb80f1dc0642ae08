"""A/B of the ct x pt inner product with 3 and 4 queries side by side over library variants (lib/variants/libhe_amd_NAME.so): python bench_tools/ab_queries.py [NAME ...]"""
import json, os, subprocess, sys
ROOT="/root/repo"
code='''
import sys
sys.path[:0]=["%s/swift-homomorphic-encryption_amd","%s/bench_tools"]
import torch, heamd, path_bench as pb
out=[]
for q in (2,3,4):
    r=pb.config5_inner_product(torch, heamd, count=256, columns=64, queries=q)
    out.append(round(r["ct_pt_mac_per_s"]/1e6,2))
print(out)
''' % (ROOT, ROOT)
for rnd in range(2):
    for name in ("production",) + tuple(sys.argv[1:]):
        env=dict(os.environ)
        if name!="production": env["HEAMD_LIBRARY"]=f"{ROOT}/swift-homomorphic-encryption_amd/lib/variants/libhe_amd_{name}.so"
        r=subprocess.run([sys.executable,"-c",code],env=env,capture_output=True,text=True)
        print(rnd,name,r.stdout.strip().splitlines()[-1] if r.returncode==0 else r.stderr[-300:])
