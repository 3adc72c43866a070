import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import quadrotor_landing_amd as qla, oracle
from util import rand_states, rand_imu, meas_near
import test_gpu_compact as tc
class MP:
    def setenv(self,k,v): os.environ[k]=v
    def delenv(self,k,raising=False): os.environ.pop(k,None)
B=1000
for direct in (0,1):
  for what in ("predict","step","update","step_none"):
    rng=np.random.default_rng(321)
    x,P=rand_states(rng,B,9,cov_scale=0.3); x[:,10:16]=0
    out=[]
    for compact in (False,True):
        ekf=tc._handle(B,"f64",MP(),compact,direct_orien_method=direct)
        ekf.set_state(x,P)
        r2=np.random.default_rng(5)
        u=rand_imu(r2,B)
        z=meas_near(r2, oracle.make_params(**dict(tc.NOBIAS, direct_orien_method=direct)), x)
        if what=="predict": ekf.step(u,None,None)
        elif what=="step": ekf.step(u,z,np.ones(B,np.uint8))
        elif what=="step_none": ekf.step(u,z,np.zeros(B,np.uint8))
        else: ekf.update(z,np.ones(B,np.uint8))
        out.append(ekf.get_state()); ekf.close()
    (xf,Pf),(xc,Pc)=out
    print(f"direct={direct} {what}: x max diff {np.abs(xf-xc).max():.3e}  P max diff {np.abs(Pf-Pc).max():.3e}")
