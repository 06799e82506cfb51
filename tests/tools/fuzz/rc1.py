"""One configuration against the real x264_rc_analyse_slice run by the harness on every leaving frame (helper of fuzz3.py)."""
import os, sys
sys.path.insert(0,os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..','..'))
import numpy as np
from oracle import refharness
from tests.oracle_backend import OracleBackend
from x264_amd import lib
from x264_amd.synth import make_clip
def nearest_cells(idx, typ):
    """(b-p0, p1-b) of every output from the coded order: nearest earlier-coded references before / after in display order"""
    cells=[]; refs=[]
    for f,t in zip(idx,typ):
        if t in (4,5):
            past=max(r for r in refs if r<f); fut=min(r for r in refs if r>f)
            cells.append((f-past,fut-f))
        elif t in (1,2): cells.append((0,0))
        else:
            past=max(r for r in refs if r<f); cells.append((f-past,0))
        if t!=5: refs.append(f)
    return cells
def run(preset,opts,over,W=176,H=144,nf=50,ckw=dict(seed=3,scene_cuts=(21,),pan=(3,1)),paced=True,verbose=True):
    frames=make_clip(W,H,nf,**ckw)
    r=refharness.Ref(W,H,preset,opts=opts)
    ref0=r.lookahead_run(frames); r.close()
    cells=nearest_cells(list(ref0["idx"]),list(ref0["type"]))
    r=refharness.Ref(W,H,preset,opts=opts)
    ref=r.lookahead_run(frames,with_qp_offsets=True,rc_cells=np.array(cells,np.int32)); rc=r.cfg; r.close()
    assert np.array_equal(ref["idx"],ref0["idx"])
    cfg=lib.la_config(W,H,preset,**over)
    be=OracleBackend(cfg)
    l=lib.Lookahead(cfg,backend=be.struct,max_frames=nf+4)
    outs=l.run(frames,qp_offsets=True,vbv=True,paced=paced); l.close()
    ok=[o.frame for o in outs]==list(ref["idx"]) and [o.type for o in outs]==list(ref["type"])
    mbh=(H+15)//16
    for k,o in enumerate(outs):
        if not ok: break
        if o.own_cell!=cells[k]: print("cell diff frame",o.frame,o.type,o.own_cell,cells[k]); ok=False; break
        if cfg["rc_is_cqp"] or (o.type in (4,5) and not cfg['vbv']):
            if o.type in (4,5) and not cfg['vbv'] and o.rc_satd!=-1: print('B satd should be -1'); ok=False; break
            continue
        want=ref["rc"][k]
        if o.rc_satd!=want[0]: print("satd diff frame",o.frame,o.type,o.rc_satd,want[0]); ok=False; break
        if cfg["vbv"]:
            if not np.array_equal(o.row_satds,want[1:1+mbh]): print("rows diff",o.frame,o.type,o.row_satds,want[1:1+mbh]); ok=False; break
            if o.type not in (1,2) and not np.array_equal(o.row_satds_intra,want[1+mbh:]): print("intra rows diff",o.frame,o.type,o.row_satds_intra,want[1+mbh:]); ok=False; break
    if verbose: print("OK" if ok else "BAD")
    return ok
if __name__=="__main__":
    run("medium","",{})
    run("medium","bitrate=500,vbv-bufsize=300,vbv-maxrate=600",dict(bitrate=500,vbv_bufsize=300,vbv_maxrate=600))
    run("medium","vbv-bufsize=300,vbv-maxrate=600,mbtree=0,b-pyramid=none",dict(vbv_bufsize=300,vbv_maxrate=600,mb_tree=0,b_pyramid=0))
    run("veryslow","",{})
    run("superfast","",{})
