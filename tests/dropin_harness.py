"""Container-only harness (needs /root/reference): runs the reference's OWN entry scripts' construction path with only
PYTHONPATH changed -- `PYTHONPATH=<repo>/ucnerf_amd/compat/dropin:<repo>` -- and reports which classes they got.

    python -B tests/dropin_harness.py train.py|render.py|eval.py

The script is executed unmodified from /root/reference/nerf (runpy, working directory and sys.path[0] as `python train.py`
sets them); its `main()` runs from `configs.load_config()` (gin files + bindings from the command-line flags) through
`accelerate.Accelerator()` to `models.Model(config=config)`; the harness records the instance the script built and stops at the next call
(`train_utils.create_optimizer` / `datasets.load_dataset`: datasets need the Waymo files).  Third-party packages the image lacks are stubbed HERE, in
the harness (test infrastructure): gin by tests/stubs/mini_gin.py (functional: bindings are parsed and injected), absl
flags / app, cv2, lpips, skimage, rawpy, nuscenes, pyquaternion, pycolmap, tensorboardX, mediapy, imageio, trimesh,
pymeshlab, torch_scatter.
Nothing under /root/reference is written (bytecode off, exp_name points at a temp directory)."""
import json
import os
import runpy
import sys
import tempfile
import types

REF = '/root/reference/nerf'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _mod(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import accelerate  # noqa: F401  (the real one; it probes optional trackers by name while importing)
    sys.path.insert(0, os.path.join(HERE, 'stubs'))
    import mini_gin
    sys.modules['gin'] = mini_gin
    flags = _mod('absl.flags', FLAGS=types.SimpleNamespace(gin_configs=None, gin_bindings=None),
                 DEFINE_string=lambda *a, **k: None, DEFINE_multi_string=lambda *a, **k: None,
                 DEFINE_integer=lambda *a, **k: None, DEFINE_bool=lambda *a, **k: None, DEFINE_float=lambda *a, **k: None)
    _mod('absl', flags=flags, app=_mod('absl.app', run=lambda main: main(sys.argv)), logging=_mod('absl.logging'))
    _mod('cv2')
    _mod('lpips', LPIPS=lambda net=None: None)
    _mod('rawpy')
    _mod('skimage', metrics=_mod('skimage.metrics', structural_similarity=None, peak_signal_noise_ratio=None),
         measure=_mod('skimage.measure', marching_cubes=None))
    _mod('nuscenes', nuscenes=_mod('nuscenes.nuscenes', NuScenes=object))
    _mod('pyquaternion', Quaternion=object)
    _mod('pycolmap', SceneManager=object)
    _mod('tensorboardX', SummaryWriter=object)
    _mod('mediapy')
    _mod('imageio')
    _mod('trimesh')
    _mod('pymeshlab')
    _mod('torch_scatter', segment_coo=None)
    sys.path.pop(0)


class Grabbed(Exception):
    pass


def main():
    script = sys.argv[1]
    tmp = tempfile.mkdtemp(prefix='ucn_dropin_')
    install_stubs()
    os.chdir(REF)
    sys.path.insert(0, REF)                                   # what `python train.py` puts at sys.path[0]
    ns = runpy.run_path(os.path.join(REF, script), run_name='ucn_dropin_script')      # top level: imports + flag definitions
    from absl import flags
    flags.FLAGS.gin_configs = ['configs/waymo.gin']
    flags.FLAGS.gin_bindings = [f"Config.exp_name = '{tmp}'", f"Config.checkpoint_dir = '{tmp}/ckpt'",
                                "Config.model_sky = True", "Config.brightness_correction = True",       # scripts/train_waymo.sh:11-12
                                "NerfMLP.grid_log2_hashmap_size = 12", "PropMLP.grid_log2_hashmap_size = 12"]
    import internal
    import internal.models as M
    import internal.train_utils as TU
    import internal.stepfun as SF
    import internal.configs as CF
    got = {}

    def stop(*a, **k):
        raise Grabbed()
    real_model = M.Model

    def recording_model(*a, **k):                             # the scripts call `models.Model(config=config)` through the module
        got['model'] = real_model(*a, **k)
        return got['model']
    M.Model = recording_model
    ours_create_optimizer = TU.create_optimizer
    TU.create_optimizer = stop                                # train.py:70, right behind models.Model(config=config)
    import internal.datasets as DS
    DS.load_dataset = stop                                    # render.py:107 / eval.py:85: the next call there (needs the Waymo files)
    try:
        ns['main']([])
    except Grabbed:
        pass
    finally:
        TU.create_optimizer = ours_create_optimizer
        M.Model = real_model
    model = got['model']
    import ucnerf_amd.internal.models as OURS
    import ucnerf_amd.internal.train_utils as OTU
    opt, lr_fn = TU.create_optimizer(model.config, model)
    report = dict(
        script=script,
        internal_init=internal.__file__, models_file=M.__file__, train_utils_file=TU.__file__, stepfun_file=SF.__file__,
        configs_file=CF.__file__,
        model_class=f"{type(model).__module__}.{type(model).__qualname__}",
        model_is_ours=type(model) is OURS.Model, nerf_is_ours=type(model.nerf_mlp) is OURS.NerfMLP,
        prop_is_ours=type(model.prop_mlp_0) is OURS.PropMLP,
        render_image_is_ours=M.render_image is OURS.render_image,
        bound=dict(num_levels=model.num_levels, num_prop_samples=model.num_prop_samples, num_nerf_samples=model.num_nerf_samples,
                   opaque_background=model.opaque_background, prop_disable_rgb=model.prop_mlp_0.disable_rgb,
                   prop_disable_density_normals=model.prop_mlp_0.disable_density_normals,
                   nerf_disable_density_normals=model.nerf_mlp.disable_density_normals, nerf_disable_rgb=model.nerf_mlp.disable_rgb,
                   nerf_log2_hashmap=model.nerf_mlp.grid_log2_hashmap_size, nerf_max_deg_point=getattr(model.nerf_mlp, 'max_deg_point', None),
                   nerf_table_rows=int(model.nerf_mlp.encoder.embeddings.shape[0])),
        config=dict(type=f"{type(model.config).__module__}.{type(model.config).__qualname__}", near=model.config.near,
                    far=model.config.far, batch_size=model.config.batch_size, model_sky=model.config.model_sky,
                    brightness_correction=model.config.brightness_correction),
        has_sky=hasattr(model, 'skynerf'), has_brightness=hasattr(model, 'brightness_corr'),
        state_dict_keys=sorted(model.state_dict().keys()),
        losses_are_ours=all(getattr(TU, n) is getattr(OTU, n) for n in
                            ('compute_data_loss', 'anti_interlevel_loss', 'distortion_loss', 'hash_decay_loss', 'sky_loss',
                             'transformIdentityLoss', 'clip_gradients', 'create_optimizer')),
        upstream_helpers_present=all(hasattr(TU, n) for n in ('tree_len', 'GradientScaler', 'img_warping', 'orientation_loss',
                                                             'interlevel_loss', 'opacity_loss')),
        optimizer=f"{type(opt).__module__}.{type(opt).__qualname__}", lr0=float(lr_fn(0)),
        n_params=sum(p.numel() for p in model.parameters()),
    )
    print("DROPIN_REPORT " + json.dumps(report))


if __name__ == '__main__':
    main()
