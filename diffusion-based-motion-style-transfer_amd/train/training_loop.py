"""Counterpart of the reference's `train/training_loop.py:42-348` (`TrainInpaintingLoop`): same constructor,
attributes and methods (`run_loop`, `run_step`, `forward_backward`, `_anneal_lr`, `log_step`, `ckpt_file_name`,
`save`), same step / epoch / save arithmetic, same checkpoint files (`model{step:09d}.pt` with the 96 trainable
tensors, `opt{step:09d}.pt` in torch.optim.AdamW's state layout), so a reference fine-tune resumes from ours and
vice versa.

What runs where: the objective (`diffusion.few_shot_style_finetune_losses`) drives the native training node for every
pass through the encoder stacks; the optimizer step and the logged norms are one native launch
(`optim.FusedAdamW` + `fp16_util.MixedPrecisionTrainer.optimize`); everything else here is host control flow.
Data-parallel use: pass `reducer=finetune_dp.LayerBucketReducer(model)`; its `finish()` runs between backward and
the optimizer step (the reference is single-device: `self.global_batch = self.batch_size # * dist.get_world_size()`)."""
import functools
import os

import torch

from ..diffusion import logger
from ..diffusion.fp16_util import MixedPrecisionTrainer
from ..diffusion.resample import LossAwareSampler, create_named_schedule_sampler
from ..optim import FusedAdamW


def parse_resume_step_from_filename(filename):
    """path/to/modelNNNNNNNNN.pt -> NNNNNNNNN (:39-52)."""
    split = filename.split("model")
    if len(split) < 2:
        return 0
    try:
        return int(split[-1].split(".")[0])
    except ValueError:
        return 0


def find_resume_checkpoint(save_dir, mode='model'):
    files = [f for f in os.listdir(save_dir) if f.endswith('.pt') and f.startswith(mode)]
    steps = [int(f[len(mode):len(mode) + 9]) for f in files]
    return os.path.join(save_dir, f"{mode}{sorted(steps)[-1]:09d}.pt")


def log_loss_dict_style(diffusion, ts, losses):
    for key, values in losses.items():
        logger.logkv_mean(key, values.mean().item() if key != 'loss' else values.item())


def log_loss_dict(diffusion, ts, losses):
    for key, values in losses.items():
        logger.logkv_mean(key, values.mean().item())
        for sub_t, sub_loss in zip(ts.cpu().numpy(), values.detach().cpu().numpy()):
            logger.logkv_mean(f"{key}_q{int(4 * sub_t / diffusion.num_timesteps)}", sub_loss)


class TrainInpaintingLoop:
    # attributes the reference's constructor copies from `args` one by one (:44-75), and the ones only a style fine-tune run has
    _FROM_ARGS = ("dataset", "batch_size", "lr", "log_interval", "save_interval", "resume_checkpoint", "weight_decay",
                  "lr_anneal_steps", "num_steps", "save_dir", "overwrite")
    _FINETUNE_ARGS = ("style_finetune", "semantic_guidance", "skip_steps")

    def __init__(self, args, train_platform, model, data, diffusion=None, style_data=None, reducer=None):
        self.args, self.train_platform, self.model, self.data, self.style_data = args, train_platform, model, data, style_data
        self.diffusion, self.reducer = diffusion, reducer
        for name in self._FROM_ARGS:
            setattr(self, name, getattr(args, name))
        finetune = hasattr(args, "style_finetune")            # the reference keys all three on `style_finetune` being present
        for name in self._FINETUNE_ARGS:
            setattr(self, name, getattr(args, name, 0) if (finetune or name == "skip_steps") else 0)
        self.style_example = hasattr(args, "skip_steps")
        self.microbatch = self.global_batch = self.batch_size  # no gradient accumulation; single-device batch (:73)
        self.use_fp16 = self.use_ddp = False
        self.step = self.resume_step = 0
        self.num_epochs = self.num_steps // len(self.data) + 1
        self.sync_cuda = torch.cuda.is_available()
        self.device = next(model.parameters()).device
        self._load_and_sync_parameters()
        self.mp_trainer = MixedPrecisionTrainer(model=self.model, use_fp16=False)
        if diffusion is not None:
            self.schedule_sampler_type = 'uniform'
            self.schedule_sampler = create_named_schedule_sampler(self.schedule_sampler_type, diffusion)
        self.opt = FusedAdamW(self.mp_trainer.master_params, lr=self.lr, weight_decay=self.weight_decay)
        if self.resume_step:
            self._load_optimizer_state()

    # ------------------------------------------------------------------------------ resume (:108-141)
    def _load_and_sync_parameters(self):
        rc = self.resume_checkpoint
        resume_checkpoint = find_resume_checkpoint(rc, 'model') if rc and os.path.isdir(rc) else rc
        if resume_checkpoint:
            self.resume_step = parse_resume_step_from_filename(resume_checkpoint)
            logger.log(f"loading model from checkpoint: {resume_checkpoint}...")
            missing, unexpected = self.model.load_state_dict(torch.load(resume_checkpoint, map_location=self.device), strict=False)
            assert len(unexpected) == 0
            assert all(k.startswith("motion_enc.") for k in missing)

    def _load_optimizer_state(self):
        rc = self.resume_checkpoint
        main_checkpoint = find_resume_checkpoint(rc, 'opt') if os.path.isdir(rc) else rc
        opt_checkpoint = os.path.join(os.path.dirname(main_checkpoint), f"opt{self.resume_step:09}.pt")
        if os.path.exists(opt_checkpoint):
            logger.log(f"loading optimizer state from checkpoint: {opt_checkpoint}")
            try:
                self.opt.load_state_dict(torch.load(opt_checkpoint, map_location=self.device))
            except Exception:
                pass                                   # reference swallows a mismatching state (:137-140)

    # ------------------------------------------------------------------------------ loop (:143-199)
    def _to_device(self, cond):
        cond['y'] = {k: v.to(self.device) if torch.is_tensor(v) else v for k, v in cond['y'].items()}
        return cond

    def _annealing_done(self):
        return not (not self.lr_anneal_steps or self.step + self.resume_step < self.lr_anneal_steps)

    def _next_style_example(self, it):
        """One (content clip, style conditioning) pair per epoch, cycling over `style_data` (:146-156)."""
        try:
            content, cond = next(it)
        except StopIteration:
            it = iter(self.style_data)
            content, cond = next(it)
        return it, content.to(self.device), self._to_device(cond)

    def _report(self):
        """Console line for the loss, every other logged scalar to the train platform (:167-176): step counters and the
        per-quartile `*_q*` entries stay in the logger only."""
        shown = self.step + self.resume_step
        for name, value in logger.get_current().name2val.items():
            if name == 'loss':
                print('step[{}]: loss[{:0.5f}]'.format(shown, value))
            if name in ('step', 'samples') or '_q' in name:
                continue
            self.train_platform.report_scalar(name=name, value=value, iteration=self.step, group_name='Loss')

    def run_loop(self):
        style_it = iter(self.style_data) if self.style_finetune else None
        for epoch in range(self.num_epochs):
            print(f'Starting epoch {epoch}')
            content_motion = cond_style = None
            if self.style_finetune:
                style_it, content_motion, cond_style = self._next_style_example(style_it)
            for motion, cond in self.data:
                if self._annealing_done():
                    break
                self.run_step(motion.to(self.device), self._to_device(cond), content_motion, cond_style)
                if self.step % self.log_interval == 0:
                    self._report()
                if self.step % self.save_interval == 0:
                    self.save()
                    if os.environ.get("DIFFUSION_TRAINING_TEST", "") and self.step > 0:
                        return
                self.step += 1
            if self._annealing_done():
                break
        if (self.step - 1) % self.save_interval != 0:          # the final save is named after the post-increment step (:198-199)
            self.save()

    def run_step(self, batch, cond, style_batch=None, style_cond=None):
        self.forward_backward(batch, cond, style_batch, style_cond)
        if self.reducer is not None:
            self.reducer.finish()
        self.mp_trainer.optimize(self.opt)
        self._anneal_lr()
        self.log_step()

    # ------------------------------------------------------------------------------ forward / backward (:229-296)
    def forward_backward(self, batch, cond, style_batch, style_cond):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.mp_trainer.zero_grad()
        assert self.diffusion is not None
        a = self.args
        if self.style_finetune:
            if a.use_ddim:
                rng = range(int((a.diffusion_steps - a.skip_steps) / a.diffusion_steps * 20))
            else:
                rng = range(a.diffusion_steps - a.skip_steps)
            t, weights = self.schedule_sampler.sample(batch.shape[0], self.device, rng)
        else:
            t, weights = self.schedule_sampler.sample(batch.shape[0], self.device)
        if not self.style_finetune:
            raise NotImplementedError("the shipped script only fine-tunes (--style_finetune 1, utils/parser_util.py:189)")
        compute_losses = functools.partial(
            self.diffusion.few_shot_style_finetune_losses, self.model, batch, t, style_batch,
            style_cond["y"]["inpainted_motion"], skip_steps=a.skip_steps, model_kwargs=style_cond, model_t2m_kwargs=cond,
            semantic_guidance=self.semantic_guidance, use_ddim=a.use_ddim, Ls=a.Ls)
        losses = compute_losses()
        if isinstance(self.schedule_sampler, LossAwareSampler):
            self.schedule_sampler.update_with_local_losses(t, losses["loss"].detach())
        if style_batch is None:
            loss = (losses["loss"] * weights).mean()
            log_loss_dict(self.diffusion, t, {k: v * weights for k, v in losses.items()})
        else:
            loss = losses["loss"]
            log_loss_dict_style(self.diffusion, t, losses)
        self.mp_trainer.backward(loss)

    def _anneal_lr(self):
        if not self.lr_anneal_steps:
            return
        frac_done = (self.step + self.resume_step) / self.lr_anneal_steps
        lr = self.lr * (1 - frac_done)
        for param_group in self.opt.param_groups:
            param_group["lr"] = lr

    def log_step(self):
        logger.logkv("step", self.step + self.resume_step)
        logger.logkv("samples", (self.step + self.resume_step + 1) * self.global_batch)

    # ------------------------------------------------------------------------------ checkpoints (:305-348)
    def ckpt_file_name(self):
        return f"model{(self.step + self.resume_step):09d}.pt"

    def save(self):
        # data-parallel (reducer present): replicas are identical, one rank writes, the others wait for the files
        import torch.distributed as dist
        multi = self.reducer is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi and dist.get_rank() != 0:
            dist.barrier()
            return
        state_dict = self.mp_trainer.master_params_to_state_dict(self.mp_trainer.master_params)
        drop = ('motion_enc.', 'clip_model.') if self.dataset != 'humanml' else ('controlmdm.', 'clip_model.')
        state_dict = {k: v for k, v in state_dict.items() if not k.startswith(drop)}
        logger.log("saving model...")
        os.makedirs(self.save_dir, exist_ok=True)
        with open(os.path.join(self.save_dir, self.ckpt_file_name()), "wb") as f:
            torch.save(state_dict, f)
        with open(os.path.join(self.save_dir, f"opt{(self.step + self.resume_step):09d}.pt"), "wb") as f:
            torch.save(self.opt.state_dict(), f)
        if multi:
            dist.barrier()
