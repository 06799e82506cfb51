"""VBV-centred fuzz: tests/tools/fuzz/fuzz3.py <seed> <n>.  Every configuration goes through vbv1.run (planned types / costs, row sums)
and rc1.run (the real x264_rc_analyse_slice on every leaving frame)."""
import os, sys
sys.path.insert(0,os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..','..'))
import numpy as np
import vbv1, rc1
rng=np.random.default_rng(int(sys.argv[1]))
bad=0
for t in range(int(sys.argv[2])):
    W,H=[(96,80),(100,70),(48,32),(176,144),(64,48),(128,272)][int(rng.integers(0,6))]
    bframes=int(rng.choice([0,1,2,3,5,8])); b_adapt=int(rng.integers(0,3)); pyr=int(rng.integers(0,3))
    keyint=int(rng.choice([8,24,60,250])); sc=int(rng.choice([0,40,80])); la=int(rng.choice([0,5,20,40,60]))
    wp=int(rng.integers(0,3)); og=int(rng.integers(0,2)); aqm=int(rng.integers(0,4)); mbt=int(rng.integers(0,3)>0)
    buf=int(rng.choice([20,300,2000])); mr=int(rng.choice([0,200,1000])); br=int(rng.choice([0,0,400]))
    preset=str(rng.choice(["medium","fast","faster","veryfast","superfast"]))
    if sc==0 and b_adapt==0 and mbt: sc=40
    nf=int(rng.integers(20,70))
    opts="bframes=%d,b-adapt=%d,b-pyramid=%s,keyint=%d,scenecut=%d,rc-lookahead=%d,weightp=%d,open-gop=%d,aq-mode=%d,mbtree=%d,vbv-bufsize=%d,vbv-maxrate=%d"%(bframes,b_adapt,["none","strict","normal"][pyr],keyint,sc,la,wp,og,aqm,mbt,buf,mr)
    over=dict(bframes=bframes,b_adapt=b_adapt,b_pyramid=pyr,keyint_max=keyint,scenecut=sc,rc_lookahead=la,weightp=wp,open_gop=og,aq_mode=aqm,mb_tree=mbt,vbv_bufsize=buf,vbv_maxrate=mr)
    if br: opts+=",bitrate=%d"%br; over["bitrate"]=br
    ckw=dict(seed=int(rng.integers(0,1000)), scene_cuts=tuple(sorted(int(x) for x in rng.integers(3,nf,size=int(rng.integers(0,3))))), pan=(int(rng.integers(0,6)),int(rng.integers(0,4))))
    if rng.integers(0,2): ckw["fade"]=(int(rng.integers(2,nf-12)),10,float(rng.choice([0.6,1.5])),int(rng.integers(-20,20)))
    print("TRY",preset,W,H,opts,nf,ckw,flush=True)
    try:
        ok=vbv1.run(preset,opts,over,W,H,nf,ckw,paced=bool(rng.integers(0,2)),verbose=False) and rc1.run(preset,opts,over,W,H,nf,ckw,paced=bool(rng.integers(0,2)),verbose=False)
    except Exception as e:
        print("EXC",repr(e)); ok=False
    print("OK" if ok else "BAD",flush=True)
    bad+=not ok
print("bad",bad)
