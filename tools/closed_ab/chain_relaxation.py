"""CPU only: how fast does the ERRHMM state chain forget its start state?  Jacobi relaxation of state[i] = T[state[i-1]][x[i]]
over 64 columns (every column from the current guess of its predecessor, repeated until nothing changes) on the shipped
models' class tables: sweeps needed = how far a wrong start state still matters.  usage: python tools/closed_ab/chain_relaxation.py"""
import sys
sys.path[:0]=['tests','tests/golden','.']
import numpy as np, harness
import pbsim3_amd as P
rng=np.random.default_rng(1)
for model in ["ERRHMM-ONT","ERRHMM-ONT-HQ","ERRHMM-RSII","ERRHMM-SEQUEL"]:
    ctx=P.Context(P.default_params(),-1); ctx.load_errhmm(harness.model_path(model+".model")); blob=ctx.dump_table(2); ctx.close()
    ncls=27; stride=len(blob)//ncls
    out=[]
    for c in (0,10,20,24,26):
        b=blob[c*stride:(c+1)*stride]; hdr=np.frombuffer(b[:64],dtype=np.uint32); smax=int(hdr[0]); mode=int(hdr[2])
        emis_off=64+32*(smax+1); init_off=emis_off+16*(smax+1)
        T=np.frombuffer(b[init_off:init_off+1000*(smax+1)],dtype=np.uint8).reshape(smax+1,1000)  # row 0 = init
        iters=[]
        for trial in range(300):
            x=rng.integers(0,1000,64)
            # sequential truth from a random start state
            s_in=int(rng.integers(1,smax+1))
            truth=np.zeros(64,int); s=s_in
            for i in range(64): s=int(T[s][x[i]]); truth[i]=s
            # Jacobi from guess: all columns = s_in
            st=np.full(64,s_in); k=0
            while True:
                prev=np.concatenate([[s_in],st[:-1]])
                new=T[prev,x].astype(int); k+=1
                if (new==st).all(): break
                st=new
            assert (st==truth).all()
            iters.append(k)
        iters=np.array(iters)
        out.append((63+c,smax,mode,round(iters.mean(),1),int(np.percentile(iters,50)),int(np.percentile(iters,95)),int(iters.max())))
    print(model,out)
