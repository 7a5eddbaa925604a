"""deepstructuredmixtures_amd -- MI355X-native GP-expert hot path of DeepStructuredMixtures.

Host API mirrors the Julia package (buildDSMGP / fit / predict / update / train ...); the per-leaf
numerics run in libdsmgp_hip.so (hand-written HIP for gfx950) behind the C ABI of include/dsmgp_hip.h.
"""
from .kernels import IsoSE, ArdSE, IsoLinear, ConstMean, KernelFunction
from .model import (DSMGP, PoE, gPoE, rBCM, GaussianProcess, build, buildDSMGP, buildPoE, buildBCM, fit,
                    fit_naive, predict, prediction, update_cholesky, update, infer, mll, mll_table,
                    reset_weights, getparams, setparams, mse, sse, mae, sae, nlpd, scores, updategradients, grad_mll, train, ADAM, RMSProp,
                    resident_test, finetune)
from .tree import get_leaves, get_overlap, share_schedule, route
from .datagen import regression_data

# reference (Julia) spelling -> function here
JULIA_NAMES = {"fit!": fit, "fit_naive!": fit_naive, "update!": update, "infer!": infer,
               "update_cholesky!": update_cholesky, "setparams!": setparams, "reset_weights!": reset_weights, "updategradients!": updategradients,
               "train!": train, "∇mll!": grad_mll, "finetune!": finetune}
