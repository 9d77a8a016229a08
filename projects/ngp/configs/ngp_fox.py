# Instant-NGP on the fox scene: fp16 hash grid + fused fp16-MFMA MLP, cone stepping (the reference's projects/ngp/configs/ngp_fox.py values).
_base_ = 'ngp_base.py'
dataset_dir = 'data/fox'
dataset = dict(
    _cover_=True,
    train=dict(type='NerfDataset', root_dir=dataset_dir, batch_size=4096, mode='train'),
    test=dict(type='NerfDataset', root_dir=dataset_dir, batch_size=4096, mode='test', preload_shuffle=False),
)
exp_name = "fox"
const_dt = False
fp16 = True
