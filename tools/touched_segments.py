"""How many 256-pixel row segments of the benchmark panorama hold terrain: the conversion reads
exactly those (every fragment lands on a pixel that ends up terrain, so 'something was drawn
here' = 'a terrain pixel is here').  Prints the fraction; tools/collect_pmc.sh hands it to
collect_pmc.py, which checks the gfx950 FETCH_SIZE factor on the conversion's known reads."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hzutil, horizonator_amd
R, W, H = 4200, 16000, 4000
h = horizonator_amd.horizonator(hzutil.VIEW_LAT, hzutil.VIEW_LON, W, H, dir_dems=hzutil.dem_dir_for(hzutil.VIEW_LAT, hzutil.VIEW_LON, R), render_radius_cells=R)
h.set_view(-180, 180, zfar=float(sys.argv[1]) if len(sys.argv) > 1 else 600000.0)
img = np.zeros((H, W, 3), np.uint8); rng = np.zeros((H, W), np.float32)
h.render_into(img, rng)
h.close()
nseg = (W + 255)//256
pad = np.zeros((H, nseg*256), bool); pad[:, :W] = rng >= 0
seg = pad.reshape(H, nseg, 256).any(axis=2)
# bytes the conversion reads: 8 per pixel of a touched segment (the last segment of a row is shorter) + one flag byte per wave
widths = np.minimum(256, W - 256*np.arange(nseg))
read = float((seg*widths[None, :]).sum())*8 + H*nseg
print("%.6f %.0f %.6f" % (seg.mean(), read, float((rng >= 0).mean())))
