"""Drop-in counterpart of the reference's `utils/model_util.py`: the factories the two scripts call
(creat_serval_diffusion :26-30, creat_ddpm_ddim_diffusion :33-37, create_gaussian_diffusion
:170-213), the checkpoint loaders (:9-23) and the args -> constructor-kwargs map (:108-167).
Host glue only: nothing here is accelerated, it wires the engine-backed classes together."""
from ..diffusion import gaussian_diffusion as gd
from ..diffusion.inpainting_gaussian_diffusion import InpaintingGaussianDiffusion
from ..diffusion.respace import SpacedDiffusion, space_timesteps
from ..model.mdm_forstyledataset import StyleDiffusion


def _load_allowing(model, state_dict, prefixes):
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    assert len(unexpected) == 0
    assert all(k.startswith(prefixes) for k in missing)


def load_model_wo_moenc(model, state_dict):
    _load_allowing(model, state_dict, ('motion_enc.', 'input_zero.', 'output_zero.'))


def load_model_wo_controlmdm(model, state_dict):
    _load_allowing(model, state_dict, ('controlmdm.',))


def get_cond_mode(args):
    """utils/parser_util.py get_cond_mode: text for the motion datasets, action for the action sets."""
    if getattr(args, 'unconstrained', False):
        return 'no_cond'
    if args.dataset in ['kit', 'humanml', 'bandai-1_posrot', 'bandai-2_posrot', 'stylexia_posrot']:
        return 'text'
    return 'action'


_FEATS = {'humanml': 263, 'bandai-1_posrot': 190, 'bandai-2_posrot': 190, 'stylexia_posrot': 181}


def get_transfer_args(args):
    data_rep, njoints, nfeats = 'rot6d', 25, 6                 # SMPL defaults
    if args.dataset in _FEATS:
        data_rep, njoints, nfeats = 'hml_vec', _FEATS[args.dataset], 1
    return {'modeltype': '', 'njoints': njoints, 'nfeats': nfeats, 'num_actions': 1,
            'translation': True, 'pose_rep': 'rot6d', 'glob': True, 'glob_rot': True,
            'latent_dim': args.latent_dim, 'ff_size': 1024, 'num_layers': args.layers, 'num_heads': 4,
            'dropout': 0.1, 'activation': "gelu", 'data_rep': data_rep, 'cond_mode': get_cond_mode(args),
            'cond_mask_prob': args.cond_mask_prob, 'action_emb': 'tensor', 'arch': args.arch,
            'emb_trans_dec': args.emb_trans_dec, 'clip_version': 'ViT-B/32', 'dataset': args.dataset,
            'mdm_path': getattr(args, 'mdm_path', ""),
            'semantic_discriminator_path': getattr(args, 'semantic_discriminator_path', ""),
            'zero_conv': True if getattr(args, 'zero_conv', None) else None,
            'inpainting_model_path': getattr(args, 'inpainting_model_path', "")}


def create_gaussian_diffusion(args, DiffusionClass=SpacedDiffusion, timestep_respacing=''):
    steps = args.diffusion_steps
    print(f"number of diffusion-steps: {steps}")
    if not timestep_respacing:
        timestep_respacing = [steps]
    extra = {k: getattr(args, k, 0) for k in ("lambda_sty_cons", "lambda_sty_trans", "lambda_cont_pers",
                                              "lambda_cont_vel", "lambda_diff_sty")}
    return DiffusionClass(
        use_timesteps=space_timesteps(steps, timestep_respacing),
        betas=gd.get_named_beta_schedule(args.noise_schedule, steps, 1.),
        model_mean_type=gd.ModelMeanType.START_X,              # this code base always predicts x_0
        model_var_type=gd.ModelVarType.FIXED_SMALL if args.sigma_small else gd.ModelVarType.FIXED_LARGE,
        loss_type=gd.LossType.MSE, rescale_timesteps=False,
        lambda_vel=args.lambda_vel, lambda_rcxyz=args.lambda_rcxyz, lambda_fc=args.lambda_fc, **extra)


def creat_serval_diffusion(args, ModelClass=StyleDiffusion, timestep_respacing=''):
    model = ModelClass(**get_transfer_args(args))
    return (model, create_gaussian_diffusion(args, InpaintingGaussianDiffusion, timestep_respacing=timestep_respacing),
            create_gaussian_diffusion(args))


def creat_ddpm_ddim_diffusion(args, ModelClass=StyleDiffusion, timestep_respacing=''):
    model = ModelClass(**get_transfer_args(args))
    return (model, create_gaussian_diffusion(args, InpaintingGaussianDiffusion, timestep_respacing=timestep_respacing),
            create_gaussian_diffusion(args, InpaintingGaussianDiffusion))
