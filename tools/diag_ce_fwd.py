"""forward-only losses of configs[2] and CE at 150 classes (same-box A/B via NMSA_LIB_PATH)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
import bench   # noqa: E402
dev = torch.device('cuda', 0)
a = bench.secondary_losses(dev)
b = bench.secondary_ce(dev)
print(json.dumps({'lib': os.environ.get('NMSA_LIB_PATH', 'default')[-24:],
                  'four_losses_fwd': a['four_losses_fwd']['ms'], 'four_losses_fwd_bwd': a['four_losses_fwd_bwd']['ms'],
                  'ce150_fwd': b['fwd']['ms'], 'ce150_fwd_bwd': b['fwd_bwd']['ms']}))
