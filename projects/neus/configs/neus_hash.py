# (ours - BASELINE.json configs[4]: "hash encoder + SDF-to-density render path") NeuS with the HIP multiresolution hash grid under a small SDF network:
# the encoder returns d(encoding)/d(position) (kernel_grid's dy_dx branch, HashEncode.h:205-251) and the second-order terms the eikonal loss back-propagates
# through (ngp_hash_encode_bwd_input_bwd_dy / _bwd_grid).  The reference's NeuS has no such configuration (its SDF network is frequency-encoded).
_base_ = 'neus_wmask.py'
encoder = dict(
    sdf_encoder=dict(_cover_=True, type='HashEncoder'),
)
model = dict(
    sdf_network=dict(d_out=65, d_hidden=64, n_layers=2, skip_in=[], bias=0.5, scale=1.0, geometric_init=True, weight_norm=True),
    rendering_network=dict(d_feature=64, mode='idr', d_out=3, d_hidden=64, n_layers=2, weight_norm=True, squeeze_out=True),
)
optim = dict(type='Adam', lr=2e-3, eps=1e-15, betas=(0.9, 0.99))
base_exp_dir = './log/dtu_scan24/hash'
warm_up_end = 500
end_iter = 20000
