"""Helper run under /opt/conda/bin/python3.9: scikit-image 0.18.3 transform.hough_line on random edge maps with
default angles, pylinac's 40-50 degree band at 0.01 degree (planar_imaging.py:3136-3158) and a full half-degree sweep.
Build container only."""
import sys, numpy as np, warnings
warnings.filterwarnings("ignore")
from skimage import transform
rng=np.random.default_rng(5)
out={}
imgs=[(rng.random((37,53))>0.97), (rng.random((120,90))>0.995), np.eye(40,dtype=bool)]
thetas=[None, np.deg2rad(np.linspace(40,50,1001)), np.deg2rad(np.linspace(-90,90,361))]
for k,(im,th) in enumerate(zip(imgs,thetas)):
    h,a,d=transform.hough_line(im, theta=th)
    out[f'img{k}']=im; out[f'h{k}']=h; out[f'a{k}']=a; out[f'd{k}']=d
np.savez_compressed(sys.argv[1], **out)
