# fox hyper-parameters on the procedural scene bench.py uses (no dataset needed)
_base_ = 'ngp_fox.py'
dataset = dict(
    _cover_=True,
    train=dict(type='SyntheticNerfDataset', batch_size=4096, n_images=50, W=400, H=400, aabb_scale=4, mode='train'),
    test=dict(type='SyntheticNerfDataset', batch_size=4096, n_images=4, W=400, H=400, aabb_scale=4, mode='test'),
)
exp_name = "synth"
tot_train_steps = 2000
