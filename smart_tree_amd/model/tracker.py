"""Running loss averages of an epoch (reference smart_tree/model/tracker.py:6-46, without the wandb logger)."""
from __future__ import annotations

import numpy as np


class Tracker:
    def __init__(self):
        self.running_epoch_radius_loss = []
        self.running_epoch_direction_loss = []
        self.running_epoch_class_loss = []

    def update(self, loss_dict: dict):
        self.running_epoch_radius_loss.append(loss_dict["radius"].item())
        self.running_epoch_direction_loss.append(loss_dict["direction"].item())
        self.running_epoch_class_loss.append(loss_dict["class_l"].item())

    @property
    def radius_loss(self):
        return np.mean(self.running_epoch_radius_loss)

    @property
    def direction_loss(self):
        return np.mean(self.running_epoch_direction_loss)

    @property
    def class_loss(self):
        return np.mean(self.running_epoch_class_loss)

    @property
    def total_loss(self):
        return self.radius_loss + self.direction_loss + self.class_loss

    def log(self, name, epoch, logger=print):
        """tracker.py:35-46 logs to wandb; here to any callable."""
        logger({f"{name} Total Loss": self.total_loss, f"{name} Radius Loss": self.radius_loss,
                f"{name} Direction Loss": self.direction_loss, f"{name} Class Loss": self.class_loss, "epoch": epoch})
