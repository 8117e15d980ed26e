"""CPU oracle for the denoising hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package restates, on the CPU (numpy float64 for the schedule tables, torch-CPU float32 for the
tensor math, as the reference does), the algorithm of the reference's per-step diffusion path:

  oracle/schedule.py   diffusion/gaussian_diffusion.py:22-66,128-221 ; diffusion/respace.py:8-87
  oracle/diffusion.py  diffusion/gaussian_diffusion.py:223-235,267-309,311-447,532-585,644-794,
                       948-1082 ; diffusion/inpainting_gaussian_diffusion.py:6-177 ;
                       diffusion/respace.py:129-134 ; model/cfg_sampler.py:36-43
  oracle/denoiser.py   model/mdm_forstyledataset.py:387-478,592-625 and the torch
                       nn.TransformerEncoderLayer arithmetic configured at :539-546

Who may import it: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`,
and there only as the checker / the reported CPU baseline.  Nothing under the product package
(`diffusion-based-motion-style-transfer_amd/`) imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this code.

Pinning: PINNED by golden vectors generated in the authoring container by importing the reference
itself (`tests/golden/make_golden.py`; the reference has no tests or fixtures of its own, SURVEY.md
section 4).  `tests/test_oracle_golden.py` checks every function here against those vectors.
Third-party arithmetic (torch nn.TransformerEncoder, CLIP) has no reference-side tests: the
transformer is pinned through the same golden vectors (reference run under torch 2.10 CPU), CLIP is
outside the engine (its [B,512] output is an input here).
"""
