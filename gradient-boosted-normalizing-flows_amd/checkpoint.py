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

    msg = f"Loaded pre-trained {os.path.split(path)[-1]}"
    if init_with_args and getattr(args, "boosted", False):
        if args.loaded_init_component is None or args.loaded_all_trained is None:
            raise ValueError("Cannot initialize a boosted model loaded from file, intialization parameters needed.")
        model.component = args.loaded_init_component
        model.all_trained = args.loaded_all_trained
        if getattr(args, "loaded_num_components", None) is not None:
            model.num_components = args.loaded_num_components
        msg += f"  and initialized with passed argument component={model.component} and all_trained={model.all_trained}"
    else:
        msg = f"Restoring {os.path.split(path)[-1]}"
        if "component" in ckpt:
            model.component = ckpt["component"]
        if "all_trained" in ckpt:
            model.all_trained = ckpt["all_trained"]
        msg += f", component={getattr(model, 'component', None)}, all_trained={getattr(model, 'all_trained', None)}"
    model.to(args.device)
    if verbose:
        logger.info(msg)
    return ckpt
