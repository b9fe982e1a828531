import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/object-detection-yolov3_amd'); sys.path.insert(0,ROOT+'/tests')
import numpy as np, torch
from oracle import model as om
from test_gpu_kernels import _labels
from yolo3.model import YoloV3
A=[(64,384),(384,64)]; K=2
img,n,seed=96,4,11
params = om.init_params(3,2,K,seed=seed)
g=torch.Generator().manual_seed(seed)
images=torch.randn(n,3,img,img,generator=g)
gts=_labels(np.random.default_rng(seed), n, img, A, K, 3)
R={}
for dt in (torch.float32, torch.float64):
    net=om.Net(params,3,2,K,dtype=dt,requires_grad=True)
    adam=om.AdamState(net.trainable(),1e-3)
    r=om.train_step(net,adam,images.to(dt),[torch.from_numpy(x) for x in gts],(img,img,3),A,K,n,apply=False)
    R[dt]=r
yolo=YoloV3(n,[img,img,3],K,A,learning_rate=1e-3); yolo.set_weights(params)
loss=yolo.train_step((images.cuda(),[torch.from_numpy(x).cuda() for x in gts]))
print('loss',float(loss),R[torch.float32]['loss'],R[torch.float64]['loss'])
flat=[]
for sp,d in zip(yolo.specs,yolo.get_gradients()):
    flat += [d['W'],d['b']] + ([d['gamma'],d['beta']] if sp.bn else [])
for i,(gg,a,b) in enumerate(zip(flat,R[torch.float32]['grads'],R[torch.float64]['grads'])):
    a=a.numpy().astype(np.float64); b=b.numpy(); gg=gg.astype(np.float64)
    sc=np.abs(b).max(); nb=np.linalg.norm(b)
    print('%3d %-18s scale %.2e | max: noise32 %.2e gpu %.2e | l2: noise32 %.2e gpu %.2e'%(i,str(b.shape),sc,np.abs(a-b).max()/sc,np.abs(gg-b).max()/sc,np.linalg.norm(a-b)/nb,np.linalg.norm(gg-b)/nb))
