"""The count form (oracle/wgl_count.c: exact search at three schedules, and the budget -> relaxed -> prefix pipeline) against the sequential
restatement of knossos.wgl (wgl_window.c) on random crash-heavy histories: verdict and failing op.  usage: stress_count_form.py <first seed> <last seed>
(round 4: seeds 100000 .. 219999 on six processes: 420,740 checks, 64,127 invalid histories, 10.1 million crashed-call steps, 0 mismatches)."""
import sys, time, random
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, synth
from oracle import wgl
om={"kind":1,"init":N.NIL}
lo,hi=int(sys.argv[1]),int(sys.argv[2])
rng=random.Random(lo)
bad=0; tot=0; inv=0; cls=0; t0=time.time(); how={}
for seed in range(lo,hi):
    n_ops=rng.choice([10,16,24,40,60,100,160])
    procs=rng.choice([2,3,4,6,8,12])
    ev=synth.register_events(n_ops=n_ops,n_procs=procs,seed=seed,busy=rng.choice([0.2,0.5,0.8,0.95]),info=rng.choice([0.05,0.15,0.3,0.5]),
                             corrupt=rng.choice([0.0,0.0,0.3,0.6,0.9]),n_values=rng.choice([2,3,5]),read=rng.choice([0.2,1/3,0.5]),write=rng.choice([0.2,1/3,0.4]))
    ops=columns.pair_events(ev).as_dict()
    if seed%2==0:
        a=ops["a"].copy(); f=ops["f"]
        idx=[i for i in range(len(f)) if f[i]==0 and a[i]!=N.NIL and ops["ret_pos"][i]!=0xFFFFFFFF]
        if idx:
            i=rng.choice(idx); a[i]=rng.randrange(5); ops["a"]=a
    e=wgl.check(ops,om,"window",want_witness=False,max_steps=2_000_000)
    if e["valid"]==-1: continue
    inv+= e["valid"]==0
    for K,rp in ((1,64),(4,64),(1,8)):
        g=wgl.check_count(ops,om,width=K,round_pairs=rp,lookahead=bool(seed&4))
        if g is None: continue
        tot+=1; cls+=g["class_steps"]
        if g["valid"]!=e["valid"] or (e["valid"]==0 and g["fail_op"]!=e["fail_op"]):
            bad+=1; print("MISMATCH exact",seed,K,rp,e["valid"],g["valid"],e["fail_op"],g["fail_op"],flush=True)
    r=wgl.check_count_pipeline(ops,om,width=rng.choice([1,2,4]),budget=rng.choice([1,3,10,50]))
    if r is not None:
        tot+=1; how[r[4]]=how.get(r[4],0)+1
        if r[0]!=e["valid"] or (r[0]==0 and r[1]!=e["fail_op"]):
            bad+=1; print("MISMATCH pipeline",seed,r[4],e["valid"],r[0],e["fail_op"],r[1],flush=True)
print("range",lo,hi,"checks",tot,"mismatches",bad,"invalid histories",inv,"class steps",cls,how,"%.0fs"%(time.time()-t0),flush=True)
