"""can a device kernel reproduce the reference's fp32 RANSAC candidate solve (tools.py:141-154:
torch.inverse(At @ A + 1e-6) @ At @ B) bit for bit, so that d_ground could be asserted against the reference on ANY input?
The 5-term products (At @ A, inv @ At, @ B) can: torch's CPU bmm takes the naive sequential loop for these sizes and a float32
emulation matches it 300 / 300.  The inverse cannot: torch.inverse = MKL sgetrf + sgetrs, whose operation order (FMA use, reciprocal
or division, small-matrix kernels chosen per CPU type) is neither documented nor stable across hosts -- four textbook variants of
partial-pivoting LU match MKL's factors on 0 / 0 / 58 % / 72 % of near-constant-depth samples and the inverse on none, and at
cond(AtA) ~ 8e8 > 1/eps one differing rounding moves the entries of the inverse by a factor of two.  Conclusion: in
the degenerate regime the reference's value is a property of its BLAS build and host CPU; the kernel solves in fp64 and is pinned
on the reference's own candidates and on trained networks instead.   python scripts/probe_fp32_inverse.py"""
import numpy as np, torch
torch.manual_seed(0)
f32=np.float32
def mm_naive(a,b):  # sequential accumulate from 0, no fma
    n,k=a.shape; k2,m=b.shape
    out=np.zeros((n,m),f32)
    for i in range(n):
        for j in range(m):
            acc=f32(0)
            for kk in range(k):
                acc=f32(acc+f32(a[i,kk]*b[kk,j]))
            out[i,j]=acc
    return out
def fma(a,b,c):
    return f32(np.float64(a)*np.float64(b)+np.float64(c))
def lu_solve_identity(M, use_fma):
    A=M.copy(); n=3
    piv=list(range(n))
    for k in range(n):
        p=k+int(np.argmax(np.abs(A[k:,k])))
        if p!=k:
            A[[k,p]]=A[[p,k]]; piv[k],piv[p]=piv[p],piv[k]
        r=f32(1)/A[k,k]
        for i in range(k+1,n):
            A[i,k]=f32(A[i,k]*r)   # scal by reciprocal (LAPACK getf2 uses reciprocal if |pivot|>=sfmin)
        for i in range(k+1,n):
            for j in range(k+1,n):
                A[i,j]= fma(-A[i,k],A[k,j],A[i,j]) if use_fma else f32(A[i,j]-f32(A[i,k]*A[k,j]))
    I=np.eye(3,dtype=f32)[piv]
    X=np.zeros((3,3),f32)
    for c in range(3):
        y=I[:,c].copy()
        for i in range(n):
            for j in range(i):
                y[i]= fma(-A[i,j],y[j],y[i]) if use_fma else f32(y[i]-f32(A[i,j]*y[j]))
        for i in reversed(range(n)):
            for j in range(i+1,n):
                y[i]= fma(-A[i,j],y[j],y[i]) if use_fma else f32(y[i]-f32(A[i,j]*y[j]))
            y[i]=f32(y[i]/A[i,i])
        X[:,c]=y
    return X
# near-constant-depth points
ok_bmm=0; ok_inv={False:0,True:0}; N=300
for t in range(N):
    P = torch.tensor([0.5,1.6,7.0]) + 0.05*torch.randn(5,3)*torch.tensor([40.,0.02,0.02])
    Bv = P[:,1:2]; A = torch.cat([P[:,0:1],P[:,2:3],torch.ones(5,1)],-1)
    At = A.t()
    AtA_t = (At.unsqueeze(0)@A.unsqueeze(0))[0]
    AtA_e = mm_naive(At.numpy(),A.numpy())
    ok_bmm += np.array_equal(AtA_t.numpy(),AtA_e)
    M = (AtA_t+1e-6)
    inv_t = torch.inverse(M.unsqueeze(0))[0].numpy()
    for uf in (False,True):
        inv_e = lu_solve_identity(M.numpy(),uf)
        ok_inv[uf] += np.array_equal(inv_t,inv_e)
    if t<2:
        print(inv_t); print(lu_solve_identity(M.numpy(),False)); print(np.linalg.cond(M.numpy().astype(np.float64)))
print("bmm exact",ok_bmm,"/",N,"inv exact nofma",ok_inv[False],"fma",ok_inv[True])

print("---- which part differs: LU or the solve?")
def lu_only(M, use_fma, recip):
    A=M.copy(); n=3
    for k in range(n):
        p=k+int(np.argmax(np.abs(A[k:,k])))
        if p!=k: A[[k,p]]=A[[p,k]]
        for i in range(k+1,n):
            A[i,k]= f32(A[i,k]*(f32(1)/A[k,k])) if recip else f32(A[i,k]/A[k,k])
        for i in range(k+1,n):
            for j in range(k+1,n):
                A[i,j]= fma(-A[i,k],A[k,j],A[i,j]) if use_fma else f32(A[i,j]-f32(A[i,k]*A[k,j]))
    return A
cnt={}
for t in range(200):
    P = torch.tensor([0.5,1.6,7.0]) + 0.05*torch.randn(5,3)*torch.tensor([40.,0.02,0.02])
    A = torch.cat([P[:,0:1],P[:,2:3],torch.ones(5,1)],-1)
    M = ((A.t().unsqueeze(0)@A.unsqueeze(0))[0]+1e-6)
    LU,piv = torch.linalg.lu_factor(M)
    for uf in (False,True):
        for rc in (False,True):
            e = lu_only(M.numpy(),uf,rc)
            cnt[(uf,rc)] = cnt.get((uf,rc),0)+np.array_equal(LU.numpy(),e)
print(cnt)
