import sys, numpy as np, scipy.linalg
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import ppbo_oracle as orc
def slices(M, S, axis):
    # per-row (axis=1) or per-column (axis=0) scaling; 7-bit signed slices
    mx = np.max(np.abs(M), axis=axis, keepdims=True); mx[mx==0]=1
    e = np.ceil(np.log2(mx))          # |M| < 2^e
    R = M / (2.0**e)                   # in (-1,1)
    out=[]
    for s in range(S):
        R = R*128.0
        q = np.trunc(R)                # integer in [-127,127]
        out.append(q.astype(np.int64))
        R = R - q
    return out, e
for name in sys.argv[1:]:
    g=dict(np.load(f'/root/repo/tests/golden/{name}.npz'))
    X,th,m=g['X'],g['theta'],int(g['m']); kern=str(g['kernel'])
    Sinv=orc.pd_inverse(orc.gram(X,th,kern)); f=g['fMAP']
    lam=orc.lambda_dense(f,m,th[0])
    B=Sinv-lam
    LB=np.linalg.cholesky(B); R=scipy.linalg.solve_triangular(LB,np.eye(len(f)),lower=True)
    G=R@lam
    Xc=g['Xc'][:256]
    K=orc.cross_cov(X,Xc,th,kern)
    Y=G@K; s=(Y**2).sum(0); t=np.einsum('ij,ij->j',K,lam@K)
    var=th[2]**2+t+s
    sf2=th[2]**2
    print(name,'N',len(f),'max|G|',np.abs(G).max(),'s/sf2 range',(s/sf2).min(),(s/sf2).max(),'t/sf2',(t/sf2).min(),(t/sf2).max(),'var/sf2',(var/sf2).min(),(var/sf2).max(), 'ref var err', np.abs(var-g['var'][:256]).max()/sf2)
    for S in (4,5,6,7):
        Gs,eg=slices(G,S,1); Ks,ek=slices(K,S,0)
        Yh=np.zeros_like(Y)
        for a in range(S):
            for b in range(S):
                if a+b<=S-1:
                    Yh+= (Gs[a]@Ks[b]).astype(np.float64)*2.0**(-7*(a+b+2))
        Yh=Yh*(2.0**eg)*(2.0**ek)
        sh=(Yh**2).sum(0)
        print('  S',S,'max |ds|/sf2',np.abs(sh-s).max()/sf2,'products',sum(1 for a in range(S) for b in range(S) if a+b<=S-1))
