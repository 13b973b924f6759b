import sys, os
sys.path.insert(0, '/root/repo')
import tonic_amd._lib as L
L.LIBRARY_PATH = os.path.join(os.path.dirname(L.LIBRARY_PATH), 'libtonic_hip_alt.so')
exec(open('/root/repo/scripts/grad_timing.py').read())
