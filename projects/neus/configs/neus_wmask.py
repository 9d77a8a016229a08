# NeuS with mask supervision: no background model, no cosine annealing, mask loss on (the reference's projects/neus/configs/neus_wmask.py values).
_base_ = 'neus_womask.py'
render = dict(n_outside=0)          # no outside NeRF: the mask removes the background
base_exp_dir = './log/dtu_scan24/wmask'
anneal_end = 0
mask_weight = 0.1
