"""Build a tuning variant of the library next to the product one:  python tools/build_variant.py NAME -DFLAG ...
-> build_variants/NAME.so (git-ignored; travels to the GPU box).  tools/variant_stats.sh runs pass_stats with each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tropical_cyclone_risk_amd import build as b
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, 'build_variants'), exist_ok=True)
os.environ['TCR_HIPCC_FLAGS'] = ' '.join(sys.argv[2:])
b.OUT = os.path.join(root, 'build_variants', sys.argv[1] + '.so')
print(b.build(force=True, verbose=True))
