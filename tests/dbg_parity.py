import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import hzutil, oracle
LAT,LON=hzutil.VIEW_LAT,hzutil.VIEW_LON
for (R,W,H) in [(32,256,64),(300,1200,300)]:
    d=hzutil.dem_dir_for(LAT,LON,R)
    od=oracle.Dem(LAT,LON,d,radius_cells=R); m=od.mosaic()
    v=od.view(LAT,LON,W,H,-180,180)
    hip=hzutil.hip_render(m,v,W,H); orc=oracle.render(m,v,W,H)
    bad=np.argwhere(hip['index']!=orc['index'])
    print(R,W,H,"mismatch px",len(bad))
    N=2*R
    for (y,x) in bad[:10]:
        hi,oi=hip['index'][y,x],orc['index'][y,x]
        def dec(p): 
            if p<0: return None
            c=p>>1; return (c%(N-1), c//(N-1), p&1)
        print("  px",x,y,"hip",hi,dec(hi),hip['z24'][y,x],"orc",oi,dec(oi),orc['z24'][y,x])
    if len(bad):
        miss=(hip['index']<0)&(orc['index']>=0); extra=(hip['index']>=0)&(orc['index']<0)
        print("  hip missing",miss.sum(),"hip extra",extra.sum())
        prims=np.unique(orc['index'][hip['index']!=orc['index']])
        print("  distinct oracle prims at mismatches", len(prims), prims[:20])
