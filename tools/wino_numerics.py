"""fp32 numerics of the 1-D Winograd transforms used by conv64_wino.hip (F(4,3)) against direct fp32 summation and a float64
reference, on post-ReLU activations with Glorot-scale weights: max and rms error relative to the output scale."""
import numpy as np
BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=np.float64)
G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],dtype=np.float64)
AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],dtype=np.float64)
BT2 = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=np.float64)
G2 = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]); AT2=np.array([[1,1,1,0],[0,1,-1,-1]],dtype=np.float64)
rng=np.random.default_rng(0)
# 1D check exactness
d=rng.normal(size=6); g=rng.normal(size=3)
y=AT@((G@g)*(BT@d)); ref=np.array([sum(g[t]*d[i+t] for t in range(3)) for i in range(4)])
print("exact F43", np.abs(y-ref).max())
def run(m, BT,G,AT, W=48, C=64, taps=9, dtype=np.float32, xs=1.0, ws=0.03):
    n=m+2
    # emulate conv over w with 'taps' other-dim taps and C cin: y[w,co]= sum_{tap,ci,t} x[tap,w+t,ci] g[tap,t,ci,co]
    x=(rng.normal(size=(taps,W+2,C))*xs); x=np.maximum(x,0)   # post-relu activations
    g=(rng.normal(size=(taps,3,C,64))*ws)
    ref=np.zeros((W,64))
    for t in range(3): ref+=np.einsum('awc,aco->wo',x[:,t:t+W],g[:,t])
    xf=x.astype(dtype); U=np.einsum('kt,atco->akco',G,g).astype(dtype)   # weights transformed in f64 then rounded (pack kernel could do fp32)
    out=np.zeros((W,64),dtype)
    for p in range(W//m):
        dd=xf[:,p*m:p*m+n]                                  # (taps,n,C)
        V=np.einsum('kn,anc->akc',BT.astype(dtype),dd).astype(dtype)
        M=np.zeros((n,64),dtype)
        for a in range(taps):
            for k in range(n):
                M[k]+= (V[a,k].astype(dtype)@U[a,k]).astype(dtype)   # fp32 accumulate (numpy matmul f32)
        out[p*m:(p+1)*m]=(AT.astype(dtype)@M)
    direct=np.zeros((W,64),dtype)
    gf=g.astype(dtype)
    for a in range(taps):
        for t in range(3): direct+= xf[a,t:t+W]@gf[a,t]
    s=np.abs(ref).max()
    return np.abs(out-ref).max()/s, np.abs(direct-ref).max()/s, np.sqrt(((out-ref)**2).mean())/np.sqrt((ref**2).mean()), np.sqrt(((direct-ref)**2).mean())/np.sqrt((ref**2).mean())
print("F(4,3) max/max: wino %.2e direct %.2e | rms wino %.2e direct %.2e"%run(4,BT,G,AT))
print("F(2,3) max/max: wino %.2e direct %.2e | rms wino %.2e direct %.2e"%run(2,BT2,G2,AT2))
