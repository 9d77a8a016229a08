"""Config presets built in code (bench.py, smoke(), tests): the hyper-parameters of the reference's projects/ngp/configs/ngp_fox.py and
ngp_base.py with the procedural SyntheticNerfDataset in place of data/fox / data/lego (neither exists on the GPU box)."""
from .utils.config import reset_cfg


def ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda", rank=0, world_size=1, tot_train_steps=40000,
            n_rays_per_batch=4096, target_batch_size=1 << 18, exp_name="synth", log_dir="./logs", scene="spheres", **extra):
    ds = dict(type="SyntheticNerfDataset", batch_size=n_rays_per_batch, n_images=n_images, W=W, H=H, aabb_scale=aabb_scale, scene=scene)
    return reset_cfg(
        sampler=dict(type="DensityGridSampler", update_den_freq=16),
        encoder=dict(pos_encoder=dict(type="HashEncoder"), dir_encoder=dict(type="SHEncoder")),
        model=dict(type="NGPNetworks", use_fully=True),
        loss=dict(type="HuberLoss", delta=0.1),
        optim=dict(type="Adam", lr=1e-1, eps=1e-15, betas=(0.9, 0.99)),
        ema=dict(type="EMA", decay=0.95),
        expdecay=dict(type="ExpDecay", decay_start=20000, decay_interval=10000, decay_base=0.33, decay_end=None),
        dataset=dict(train=dict(ds, mode="train"), test=dict(ds, mode="test", n_images=2)),
        exp_name=exp_name, log_dir=log_dir, tot_train_steps=tot_train_steps, background_color=[0, 0, 0],
        hash_func="p0 ^ p1 * 19349663 ^ p2 * 83492791", cone_angle_constant=0.00390625, near_distance=0.2,
        n_rays_per_batch=n_rays_per_batch, n_training_steps=16, target_batch_size=target_batch_size, const_dt=const_dt,
        fp16=fp16, device=device, rank=rank, world_size=world_size, **extra)
