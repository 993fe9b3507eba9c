import sys, numpy as np
sys.path.insert(0,'/root/repo/lis-slam_amd'); sys.path.insert(0,'/root/repo/oracle')
from lisreg import synth
from scipy.spatial import cKDTree
import oracle_ctypes as oc, ctypes as C
oc.build(); L=oc.lib()
case = synth.make_case(h=32, w=900, m_points=200000, scan_seed=1000)
tgt = synth.pcl_xyz(case["tgt_surf"]); src = synth.pcl_xyz(case["src_surf"])
M = synth.pose_matrix(case["T_init"])
q = (src.astype(np.float64) @ M[:3,:3].T + M[:3,3])
d,i = cKDTree(tgt.astype(np.float64)).query(q,k=5)
ok = d[:,4]**2 < 1.0
nb = tgt[i[ok]]  # [n,5,3] float32
q = q[ok]
n = len(nb); print("queries", n)
# truth float64 LSQ
nb64 = nb.astype(np.float64)
def truth(P):
    x = np.linalg.lstsq(P, -np.ones(5), rcond=None)[0]
    ps = np.linalg.norm(x); return np.r_[x/ps, 1/ps]
T = np.array([truth(P) for P in nb64])
# oracle QR float32
X = np.zeros(3,np.float32); b = -np.ones(5,np.float32)
L.orc_lstsq5x3.argtypes=[C.c_void_p,C.c_void_p,C.c_void_p]
Q = np.zeros((n,4))
for k in range(n):
    A = np.ascontiguousarray(nb[k]); L.orc_lstsq5x3(A.ctypes.data, b.ctypes.data, X.ctypes.data)
    x = X.astype(np.float32); ps = np.float32(np.sqrt(np.float32(x[0]*x[0]+x[1]*x[1]+x[2]*x[2])))
    Q[k] = np.r_[x/ps, np.float32(1)/ps]
# centred adjugate in float32
f=np.float32
def centred(P):
    P=P.astype(f)
    c = (P[0]+P[1]+P[2]+P[3]+P[4])*f(0.2)
    D = P-c
    S00=f(0);S01=f(0);S02=f(0);S11=f(0);S12=f(0);S22=f(0)
    for j in range(5):
        S00+=D[j,0]*D[j,0];S01+=D[j,0]*D[j,1];S02+=D[j,0]*D[j,2];S11+=D[j,1]*D[j,1];S12+=D[j,1]*D[j,2];S22+=D[j,2]*D[j,2]
    C00=S11*S22-S12*S12; C01=S02*S12-S01*S22; C02=S01*S12-S02*S11; C11=S00*S22-S02*S02; C12=S01*S02-S00*S12; C22=S00*S11-S01*S01
    u0=C00*c[0]+C01*c[1]+C02*c[2]; u1=C01*c[0]+C11*c[1]+C12*c[2]; u2=C02*c[0]+C12*c[1]+C22*c[2]
    det=S00*C00+S01*C01+S02*C02
    cu=c[0]*u0+c[1]*u1+c[2]*u2
    nu=f(np.sqrt(u0*u0+u1*u1+u2*u2)); r=f(1)/nu
    return np.array([-u0*r,-u1*r,-u2*r,(f(0.2)*det+cu)*r],np.float64)
Cn = np.array([centred(P) for P in nb])
def resid(Pl):  # max |n.p+d| over neighbours and pd2 for q, evaluated in float64
    r = np.abs(np.einsum('kjc,kc->kj', nb64, Pl[:,:3]) + Pl[:,3:4]).max(1)
    pd2 = np.einsum('kc,kc->k', q, Pl[:,:3]) + Pl[:,3]
    return r, pd2
rT,pT = resid(T); rQ,pQ=resid(Q); rC,pC=resid(Cn)
for name,(r,p) in (("QR f32",(rQ,pQ)),("centred f32",(rC,pC))):
    print(name, "max|res diff vs truth|", np.abs(r-rT).max(), "p99", np.quantile(np.abs(r-rT),0.99), " pd2 diff max", np.abs(p-pT).max(), "p99", np.quantile(np.abs(p-pT),.99), "median", np.median(np.abs(p-pT)))
print("QR vs centred pd2 max", np.abs(pQ-pC).max(), "normal angle diff max", np.arccos(np.clip((Q[:,:3]*Cn[:,:3]).sum(1),-1,1)).max())
print("truth vs centred angle max", np.arccos(np.clip((T[:,:3]*Cn[:,:3]).sum(1),-1,1)).max(), "truth vs QR angle max", np.arccos(np.clip((T[:,:3]*Q[:,:3]).sum(1),-1,1)).max())
