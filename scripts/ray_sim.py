#!/usr/bin/env python3
"""CPU model of k_ray's merge / atomic load at configs[3] (no GPU needed): replays the DDA of the scan's rays in
float32 with every wavefront (64 queue-consecutive rays) walking in lockstep, and counts the lowering events —
distinct (wavefront, cell) pairs whose height goes down — for a given queue order.
   python scripts/ray_sim.py [wedges=2048] [length-class shift=4] [len|slope]
DESIGN.md section 7 f1 quotes its numbers (4.12 M events for 0.99 M cells at 2048 wedges)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sys, time
from fastdem_amd import synth
f32=np.float32
wl = synth.make("c4")
s = wl.scans[0]
p = np.stack([s["x"],s["y"],s["z"]],1).astype(np.float64)
pb = p @ wl.T_base_sensor[:3,:3].T + wl.T_base_sensor[:3,3]
rng = np.linalg.norm(p,axis=1)
ok = (pb[:,2]>=-2)&(pb[:,2]<=5)&(rng>=0.5)&(rng<=40)
pw = (pb @ wl.pose(0)[:3,:3].T + wl.pose(0)[:3,3])[ok]
o = (wl.pose(0) @ wl.T_base_sensor)[:3,3]
vox = np.floor(pw/0.05).astype(np.int64)
key=((vox[:,2]+1000)<<44)|((vox[:,1]+100000)<<22)|(vox[:,0]+100000)
_,first=np.unique(key,return_index=True)
q=pw[np.sort(first)]
q=q[q[:,2]<o[2]].astype(f32)
N=len(q); print("rays",N)
res=f32(0.05); nrows=ncols=1200
cx,cy=wl.pose(0)[0,3],wl.pose(0)[1,3]
ox=f32(cx)+f32(nrows)*res*f32(0.5); oy=f32(cy)+f32(ncols)*res*f32(0.5)
sx,sy,sz=f32(o[0]),f32(o[1]),f32(o[2])
WEDGES=int(sys.argv[1]) if len(sys.argv)>1 else 2048
LSH=int(sys.argv[2]) if len(sys.argv)>2 else 4
SORTKEY=sys.argv[3] if len(sys.argv)>3 else "len"
dx=q[:,0]-sx; dy=q[:,1]-sy
ssum=np.abs(dx)+np.abs(dy); pp=dy/ssum
a=np.where(dx>=0,np.where(dy>=0,pp,4+pp),2-pp)
wedge=np.minimum(WEDGES-1,(a*(WEDGES/4)).astype(np.int64))
ln=(ssum/res).astype(np.int64)>>LSH
if SORTKEY=="slope":
    slope=(q[:,2]-sz)/np.sqrt(dx*dx+dy*dy)   # negative; steepest first
    cls=np.argsort(np.argsort(slope))  # fine rank
    order=np.lexsort((slope,wedge))
else:
    rnd=np.random.default_rng(0).random(N)
    order=np.lexsort((rnd,ln,wedge))
q=q[order]
gr0=(ox-sx)/res; gc0=(oy-sy)/res
gr1=(ox-q[:,0])/res; gc1=(oy-q[:,1])/res
dr=gr1-gr0; dc=gc1-gc0
r=np.full(N,int(np.floor(gr0))); c=np.full(N,int(np.floor(gc0)))
step_r=np.where(dr>0,1,-1); step_c=np.where(dc>0,1,-1)
br=np.where(step_r>0,f32(r+1.0),f32(r)).astype(f32); bc=np.where(step_c>0,f32(c+1.0),f32(c)).astype(f32)
tmr=((br-gr0)/dr).astype(f32); tmc=((bc-gc0)/dc).astype(f32)
tdr=(step_r.astype(f32)/dr).astype(f32); tdc=(step_c.astype(f32)/dc).astype(f32)
dz=(q[:,2]-sz).astype(f32)
state=np.full(nrows*ncols,np.inf,f32)
alive=np.ones(N,bool)
wave=np.arange(N)//64
ev_step=0; visits=0; wave_steps=0; need_lanes=0
pairs=set()
prev_pairs=np.zeros(0,np.int64); ev_nocons=0
allpairs=[]
t0=time.time()
for s_ in range(2400):
    if not alive.any(): break
    row=tmr<tmc
    texit=np.where(row,tmr,tmc)
    h=(sz+np.minimum(texit,f32(1.0))*dz).astype(f32)
    inmap=alive&(r>=0)&(r<nrows)&(c>=0)&(c<ncols)
    cell=c*nrows+r
    visits+=inmap.sum()
    wave_steps+=len(np.unique(wave[alive]))
    idx=np.nonzero(inmap)[0]
    need=h[idx]<state[cell[idx]]
    ni=idx[need]
    need_lanes+=len(ni)
    pk=np.unique(wave[ni].astype(np.int64)*(nrows*ncols)+cell[ni])
    ev_step+=len(pk)
    ev_nocons+=len(np.setdiff1d(pk,prev_pairs,assume_unique=True))
    prev_pairs=pk
    allpairs.append(pk)
    np.minimum.at(state,cell[ni],h[ni])
    alive&=~(texit>=1.0)
    r=np.where(row,r+step_r,r); c=np.where(row,c,c+step_c)
    tmr=np.where(row,(tmr+tdr).astype(f32),tmr); tmc=np.where(row,tmc,(tmc+tdc).astype(f32))
allp=np.unique(np.concatenate(allpairs))
print(f"wedges {WEDGES} lsh {LSH} sort {SORTKEY}: visits {visits/1e6:.1f}M wave_steps {wave_steps/1e6:.2f}M need_lanes {need_lanes/1e6:.2f}M events(wave,cell,step) {ev_step/1e6:.2f}M  not-repeated-from-prev-step {ev_nocons/1e6:.2f}M distinct(wave,cell) {len(allp)/1e6:.2f}M cells {np.isfinite(state).sum()/1e6:.2f}M  {time.time()-t0:.0f}s")
