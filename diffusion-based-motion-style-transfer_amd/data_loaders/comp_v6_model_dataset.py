"""Counterpart of the reference's only large-batch consumer of `p_sample_loop`,
`data_loaders/humanml/motion_loaders/comp_v6_model_dataset.py:146-240` (`CompMDMGeneratedDataset`): walks a
text-to-motion dataloader, generates one batch of clips per step with classifier-free guidance (`y['scale']`),
repeats selected batches `mm_num_repeats` times for the multimodality metric, and serves the generated clips like a
dataset.  SURVEY section 8f-4 / BASELINE config 5 (batched sampling).

The sampling call is the native loop (the whole 1000-step CFG loop is one C call per batch); with `shard=(rank, world)`
every rank generates only its contiguous share of the batches (no collective: clips are independent) and
`all_generated()` gathers them -- the reference is single-device.  `__getitem__` needs the loader's own dataset
helpers (`w_vectorizer`, `t2m_dataset.inv_transform`, eval mean/std), exactly as the reference does."""
import numpy as np
import torch


class CompMDMGeneratedDataset(torch.utils.data.Dataset):
    def __init__(self, model, diffusion, dataloader, mm_num_samples, mm_num_repeats, max_motion_length, num_samples_limit,
                 scale=1., shard=None):
        self.dataloader = dataloader
        self.dataset = dataloader.dataset
        assert mm_num_samples < len(dataloader.dataset)
        use_ddim = False            # hard-coded in the reference (:152)
        clip_denoised = False       # hard-coded in the reference (:153)
        self.max_motion_length = max_motion_length
        sample_fn = diffusion.p_sample_loop if not use_ddim else diffusion.ddim_sample_loop
        bs = dataloader.batch_size
        real_num_batches = len(dataloader)
        if num_samples_limit is not None:
            real_num_batches = num_samples_limit // bs + 1
        print('real_num_batches', real_num_batches)
        generated_motion, mm_generated_motions = [], []
        if mm_num_samples > 0:
            mm_idxs = np.sort(np.random.choice(real_num_batches, mm_num_samples // bs + 1, replace=False))
        else:
            mm_idxs = []
        print('mm_idxs', mm_idxs)
        rank, world = shard if shard is not None else (0, 1)
        dev = next(model.parameters()).device
        model.eval()
        produced = 0                                   # clips the UNSHARDED walk would have produced so far (limit check)
        with torch.no_grad():
            for i, (motion, model_kwargs) in enumerate(dataloader):
                if num_samples_limit is not None and produced >= num_samples_limit:
                    break
                produced += bs
                if i % world != rank:                  # another rank's batch
                    continue
                tokens = [t.split('_') for t in model_kwargs['y']['tokens']]
                model_kwargs['y'] = {k: v.to(dev) if torch.is_tensor(v) else v for k, v in model_kwargs['y'].items()}
                if scale != 1.:
                    model_kwargs['y']['scale'] = torch.ones(motion.shape[0], device=dev) * scale
                is_mm = i in mm_idxs
                repeat_times = mm_num_repeats if is_mm else 1
                mm_motions = []
                for t in range(repeat_times):
                    sample = sample_fn(model, motion.shape, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                       skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None,
                                       const_noise=False)
                    clips = sample.squeeze(2).permute(0, 2, 1).cpu().numpy()          # one D2H per batch: [B, T, F]
                    lengths = model_kwargs['y']['lengths'].cpu().numpy()
                    if t == 0:
                        generated_motion += [{'motion': clips[b], 'length': lengths[b], 'caption': model_kwargs['y']['text'][b],
                                              'tokens': tokens[b], 'cap_len': len(tokens[b]), 'batch': i} for b in range(bs)]
                    if is_mm:
                        mm_motions += [{'motion': clips[b], 'length': lengths[b]} for b in range(bs)]
                if is_mm:
                    mm_generated_motions += [{'caption': model_kwargs['y']['text'][b], 'tokens': tokens[b],
                                              'cap_len': len(tokens[b]), 'mm_motions': mm_motions[b::bs], 'batch': i}
                                             for b in range(bs)]
        self.generated_motion = generated_motion
        self.mm_generated_motion = mm_generated_motions
        self.w_vectorizer = getattr(dataloader.dataset, "w_vectorizer", None)
        self.shard = (rank, world)

    def all_generated(self, group=None):
        """Every rank's clips in the unsharded order (batch index, then position in the batch)."""
        import torch.distributed as dist
        if self.shard[1] == 1:
            return self.generated_motion
        parts = [None] * self.shard[1]
        dist.all_gather_object(parts, self.generated_motion, group=group)
        merged = [d for p in parts for d in p]
        order = sorted(range(len(merged)), key=lambda k: merged[k]['batch'])          # stable: keeps in-batch order
        return [merged[k] for k in order]

    def __len__(self):
        return len(self.generated_motion)

    def __getitem__(self, item):
        data = self.generated_motion[item]
        motion, m_length, caption, tokens = data['motion'], data['length'], data['caption'], data['tokens']
        sent_len = data['cap_len']
        if getattr(self.dataset, "mode", None) == 'eval':                 # T2M evaluators expect their own normalisation
            denormed = self.dataset.t2m_dataset.inv_transform(motion)
            motion = (denormed - self.dataset.mean_for_eval) / self.dataset.std_for_eval
        pos_one_hots, word_embeddings = [], []
        for token in tokens:
            word_emb, pos_oh = self.w_vectorizer[token]
            pos_one_hots.append(pos_oh[None, :])
            word_embeddings.append(word_emb[None, :])
        return (np.concatenate(word_embeddings, axis=0), np.concatenate(pos_one_hots, axis=0), caption, sent_len, motion,
                m_length, '_'.join(tokens))
