import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for k in a.files:
    x,y=a[k],b[k]
    if k in ("du","dstats"):
        xf,yf=x.view(np.float32),y.view(np.float32)
    else:
        xf,yf=x.view(np.uint16),y.view(np.uint16)
    neq=np.nonzero(xf!=yf)[0]
    print(k, "differing elements", len(neq), "of", len(xf), "first", neq[:10].tolist())
