"""Checkpoint files in the reference's own format (utils/utilities.py:42-93) plus the side-car it forgets.

``save`` writes exactly the dictionary ``utils.utilities.save`` writes -- ``model`` (state_dict), ``optimizer``,
``scheduler``, ``all_trained``, ``component`` -- so the reference can read the file, and adds ONE extra key,
``gbnf_side_car``: the permutation indices and ActNorm ``inited`` flags that the reference keeps outside
``state_dict`` (models/layers.py:633-651, 461-471) and therefore loses on every save/restore (a restored
Glow component silently gets fresh random shuffles).

``load`` reads both kinds of file: ours (side-car installed) and the reference's own (no side-car: the parameters are
restored, and unless the caller passes the indices exported from the live reference model -- INTEGRATION.md section A --
a warning says that the permutations in this process are NOT the ones the parameters were trained with).
"""
from __future__ import annotations

import logging
import os
import warnings

import torch

logger = logging.getLogger(__name__)
SIDE_CAR_KEY = "gbnf_side_car"


def save(model, optimizer, path, scheduler=None):
    """utils/utilities.py:78-93, same keys, + the side-car."""
    ckpt = {
        "model": model.state_dict(),
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "scheduler": scheduler.state_dict() if scheduler is not None else None,
    }
    if hasattr(model, "component") and hasattr(model, "all_trained"):
        ckpt["all_trained"] = model.all_trained
        ckpt["component"] = model.component
    if hasattr(model, "permutation_state"):
        ckpt[SIDE_CAR_KEY] = model.permutation_state()
    torch.save(ckpt, path)


def load(model, optimizer, path, args, init_with_args=False, scheduler=None, verbose=True, side_car=None):
    """utils/utilities.py:42-75, same argument meaning.  ``side_car``: permutation state exported from a live reference
    model, for files written by the reference itself."""
    ckpt = torch.load(path, map_location=args.device)
    model.load_state_dict(ckpt["model"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    if scheduler is not None and ckpt.get("scheduler") is not None:
        scheduler.load_state_dict(ckpt["scheduler"])

    side_car = side_car if side_car is not None else ckpt.get(SIDE_CAR_KEY)
    if side_car is not None and hasattr(model, "load_permutation_state"):
        model.load_permutation_state(side_car)
    elif getattr(model, "component_type", None) == "glow":
        warnings.warn(
            f"{os.path.split(path)[-1]} holds no permutation indices (the reference does not checkpoint them, "
            "SURVEY.md S5): this model's shuffles are the ones drawn at construction, not the ones the parameters were "
            "trained with.  Export them from the live reference model (INTEGRATION.md section A) and pass side_car=.")
        for flow in model.flows:            # the reference's restored ActNorm layers behave as initialised ones only
            if hasattr(flow, "set_actnorm_init"):   # after set_actnorm_init(); a trained checkpoint is initialised
                flow.set_actnorm_init()

    # Boosting position: from the caller's --loaded_* arguments when asked for (the reference's `init_with_args`, used to
    # continue training a file with a different component count), else from the file.
    name = os.path.basename(path)
    override = bool(init_with_args) and bool(getattr(args, "boosted", False))
    if override:
        component = getattr(args, "loaded_init_component", None)
        all_trained = getattr(args, "loaded_all_trained", None)
        if component is None or all_trained is None:
            raise ValueError("init_with_args needs args.loaded_init_component and args.loaded_all_trained to place a boosted "
                             "model that was loaded from a file")
        model.component, model.all_trained = component, all_trained
        n_loaded = getattr(args, "loaded_num_components", None)
        if n_loaded is not None:
            model.num_components = n_loaded
        msg = f"{name}: parameters loaded, boosting position from the arguments (component {component}, all_trained {all_trained})"
    else:
        for key in ("component", "all_trained"):
            if key in ckpt:
                setattr(model, key, ckpt[key])
        msg = (f"{name}: restored, component {getattr(model, 'component', None)}, "
               f"all_trained {getattr(model, 'all_trained', None)}")
    model.to(args.device)
    if verbose:
        logger.info(msg)
    return ckpt
