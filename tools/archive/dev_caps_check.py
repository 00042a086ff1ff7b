import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import gp_oracle as orc
from andvaranaut_amd import MiGP
for kernel,d,N in [("RBF+Matern52*Matern32+RBF+Matern52",130,150),("RBF+Matern52+Matern32+RBF*Matern52+Matern32+RBF+Matern52",17,260),("Matern52*RBF",300,140)]:
    kerns=kernel.replace("*","+").split("+"); ops=[c for c in kernel if c in "+*"]
    X,y=orc.synth_problem(N,d,seed=3)
    th=orc.synth_theta(d,nkern=len(kerns),gv=1e-3); th[:len(kerns)*d]*=np.sqrt(d/2)
    gp=MiGP(X,y,kernel)
    v,g,gy,gX=gp.lml_grad_data(th)
    r,gr=orc.lml_grad(X,y,kerns,ops,th); _,gyr,gXr=orc.lml_grad_data(X,y,kerns,ops,th)
    Xn=np.random.default_rng(0).random((3,d))
    mu,var,dm,dv=gp.predict_grad(th,Xn); dmo,dvo=orc.predict_grad(X,y,Xn,kerns,ops,th)
    print(kernel[:30],d,"lml",abs(v-r)/abs(r),"g",np.abs(g-gr).max()/np.abs(gr).max(),"gX",np.abs(gX-gXr).max()/np.abs(gXr).max(),"dm",np.abs(dm-dmo).max()/np.abs(dmo).max(),"dv",np.abs(dv-dvo).max()/np.abs(dvo).max())
    gp.close()
