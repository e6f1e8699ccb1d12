import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, scenes as S, server_amd as A
from fuzz_scenes import fuzz_scene
seed, bits = int(sys.argv[1]), int(sys.argv[2])
sc = fuzz_scene(seed, bits)
m = A.Mixer(channel_id=1, gpu=0)
print(sc["width"], sc["height"], sc.get("square"), sc["field_mode"])
groups=[]; depth=0; cur=[]
for op in sc["ops"]:
    cur.append(op)
    if op[0]=="push": depth+=1
    if op[0]=="pop":
        depth-=1
        if depth==0: groups.append(cur); cur=[]
for k in range(1,len(groups)+1):
    s2=dict(sc); s2["ops"]=[o for g in groups[:k] for o in g]
    cache={}
    a=S.run_product(s2,mixer=m,planes_cache=cache); b=S.run_oracle(s2,"cpu",cache)
    d=np.abs(a.astype(int)-b.astype(int)).max(axis=2)
    ys,xs=np.nonzero(d>0)
    print(k, S.diff_stats(a,b), [(int(y),int(x),a[y,x].tolist(),b[y,x].tolist()) for y,x in list(zip(ys,xs))[:6]])
    for op in groups[k-1]:
        if op[0]=="visit": print("    visit", op[1]["fmt"], op[1]["planes"])
        elif op[0]=="push": print("   ", op)
m.close()
