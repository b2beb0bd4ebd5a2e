import sys, os, time, threading, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import herc_oracle as ho
ho.lib()
nx,ny,nz,h,dt,freq=64,64,32,1000.0/128,3.6e-4,50.0
e,l,n=ho.uniform_mesh(nx,ny,nz)
ed=np.empty((len(l),4),np.float32); ed[:]=(h,6000,3464,2700)
et,nt=ho.solver_init(l,ed,ho.face_bits(e,nx,ny,nz),len(n),dt,freq)
K=ho.compute_K(); E=len(l); N=len(n)
rng=np.random.default_rng(1); b1=rng.uniform(-1,1,(N,3))*1e-3; b2=b1*0.999
for cores in (1,8,32,64,128,256):
    st=[(b1.copy(),b2.copy()) for _ in range(cores)]
    def work(i): ho.solver_run(l,et,nt,st[i][0],st[i][1],0,3,dt,formulation=0,K=K)
    th=[threading.Thread(target=work,args=(i,)) for i in range(cores)]
    t0=time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; el=time.perf_counter()-t0
    print(cores, 'threads: %.2f M elem/s total, %.3f per thread'%(cores*E*3/el/1e6, E*3/el/1e6), flush=True)
