import numpy as np
a=np.load('gpurun_out/st_base.npy'); b=np.load('gpurun_out/st_new.npy')
for c in range(5):
    d=(a[:,c]!=b[:,c])
    print('plane',c,'mismatch',d.mean(), 'max rel', np.max(np.abs(a[:,c]-b[:,c])/np.abs(a[:,c]).clip(1e-30)))
    if d.any():
        idx=np.argwhere(d)[:5]; print(idx.tolist(), [ (a[i[0],c,i[1],i[2]], b[i[0],c,i[1],i[2]]) for i in idx])
