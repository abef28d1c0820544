"""CPU estimate for DESIGN.md section 7 (verdict item 4): of the 8x8 pixel blocks a splat of the metric workload reaches (exact ellipse-vs-rectangle test,
as block_hit in csrc/blend.hip), how many of the four 4x4 quadrants does it reach?  Prints the mean and the histogram."""
import sys, numpy as np, torch, math
sys.path.insert(0,'/root/repo')
from sugar_amd import synthetic as syn
scene, cams, bg = syn.make_config("metric", P=40000)
cam = cams[0]
# project with plain numpy (EWA): restated quickly from the oracle's formulas
W,H = cam.image_width, cam.image_height
V = cam.viewmatrix.numpy().astype(np.float64); PM = cam.projmatrix.numpy().astype(np.float64)
m = scene.means3D.numpy().astype(np.float64)
ph = np.c_[m, np.ones(len(m))] @ PM; pw = 1/(ph[:,3]+1e-7); pp = ph[:,:3]*pw[:,None]
pv = np.c_[m, np.ones(len(m))] @ V
vis = pv[:,2] > 0.2
fx = W/(2*cam.tanfovx); fy = H/(2*cam.tanfovy)
q = scene.rotations.numpy().astype(np.float64); s = scene.scales.numpy().astype(np.float64)
r,x,y,z = q.T
R = np.stack([1-2*(y*y+z*z), 2*(x*y-r*z), 2*(x*z+r*y), 2*(x*y+r*z), 1-2*(x*x+z*z), 2*(y*z-r*x), 2*(x*z-r*y), 2*(y*z+r*x), 1-2*(x*x+y*y)],1).reshape(-1,3,3)
Sig = R @ (s[:,:,None]**2 * np.transpose(R,(0,2,1)))
tx,ty,tz = pv[:,0],pv[:,1],pv[:,2]
limx=1.3*cam.tanfovx; limy=1.3*cam.tanfovy
tx = np.clip(tx/tz,-limx,limx)*tz; ty=np.clip(ty/tz,-limy,limy)*tz
J = np.zeros((len(m),2,3)); J[:,0,0]=fx/tz; J[:,0,2]=-fx*tx/tz**2; J[:,1,1]=fy/tz; J[:,1,2]=-fy*ty/tz**2
Wm = V[:3,:3].T
T = J @ Wm[None]
cov = T @ Sig @ np.transpose(T,(0,2,1))
cov[:,0,0]+=0.3; cov[:,1,1]+=0.3
det = cov[:,0,0]*cov[:,1,1]-cov[:,0,1]**2
cx = cov[:,1,1]/det; cy=-cov[:,0,1]/det; cz=cov[:,0,0]/det
px = ((pp[:,0]+1)*W-1)*0.5; py=((pp[:,1]+1)*H-1)*0.5
op = scene.opacities.numpy()[:,0].astype(np.float64)
ok = vis & (det>0) & (op>1/255) & (px>-50)&(px<W+50)&(py>-50)&(py<H+50)
idx = np.nonzero(ok)[0]
def hit(gx,gy,cx,cy,cz,tau2,x0,y0,size):
    # exact min of q over rect [x0,x0+size-1]^2 (pixel centres), as block_hit
    dxl=x0-gx; dxr=dxl+size-1; dyl=y0-gy; dyh=dyl+size-1
    inside=(dxl<=0)&(dxr>=0)&(dyl<=0)&(dyh>=0)
    det=cx*cz-cy*cy
    def edge_v(dx):  # vertical edge at dx: minimise over dy
        dys=np.clip(-cy*dx/cz,dyl,dyh); return cx*dx*dx+2*cy*dx*dys+cz*dys*dys
    def edge_h(dy):
        dxs=np.clip(-cy*dy/cx,dxl,dxr); return cx*dxs*dxs+2*cy*dxs*dy+cz*dy*dy
    qmin=np.minimum(np.minimum(edge_v(dxl),edge_v(dxr)),np.minimum(edge_h(dyl),edge_h(dyh)))
    return inside|(qmin<=tau2)
rng=np.random.default_rng(0)
tot_blocks=0; tot_quads=0; hist=np.zeros(5)
for i in idx[:6000]:
    tau2=2*math.log(255*op[i])
    # radius bound
    lam=0.5*(cov[i,0,0]+cov[i,1,1])+math.sqrt(max(0.1,(0.5*(cov[i,0,0]+cov[i,1,1]))**2-det[i]))
    rad=math.ceil(3*math.sqrt(lam))
    bx0=int(max(0,(px[i]-rad)//8)); bx1=int(min((W-1)//8,(px[i]+rad)//8)); by0=int(max(0,(py[i]-rad)//8)); by1=int(min((H-1)//8,(py[i]+rad)//8))
    if bx1<bx0 or by1<by0: continue
    X,Y=np.meshgrid(np.arange(bx0,bx1+1)*8,np.arange(by0,by1+1)*8)
    X=X.ravel().astype(float);Y=Y.ravel().astype(float)
    hb=hit(px[i],py[i],cx[i],cy[i],cz[i],tau2,X,Y,8)
    if not hb.any(): continue
    Xb=X[hb];Yb=Y[hb]
    nq=np.zeros(len(Xb))
    for ox in (0,4):
        for oy in (0,4):
            nq+=hit(px[i],py[i],cx[i],cy[i],cz[i],tau2,Xb+ox,Yb+oy,4)
    tot_blocks+=len(Xb); tot_quads+=nq.sum()
    for k in range(5): hist[k]+=(nq==k).sum()
print("block hits",tot_blocks,"mean quadrants per block hit",tot_quads/tot_blocks, "hist", hist/hist.sum())
