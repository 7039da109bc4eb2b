"""Immutable attribute namespace over the YAML configuration -- same surface as the
reference's ``sc/utils/parameter.py:42-94`` (``Parameters``, ``from_yaml``, ``get``,
``update``, ``to_dict``; attribute assignment raises ``TypeError``)."""
import yaml

from .model import AE_CLS_DICT  # noqa: F401  (re-exported like the reference does)

# optimizers reachable through Trainer.from_data (SURVEY.md finding 4); AdaBound/RAdam need
# torch_optimizer, which neither the reference's runnable set nor this image provides.
OPTIM_NAMES = ("Adam", "AdamW")


class Parameters:
    def __init__(self, parameter_dict):
        object.__setattr__(self, "_parameter_dict", parameter_dict)
        self.update(parameter_dict)

    def __setattr__(self, name, value):
        raise TypeError("Parameters object cannot be modified after instantiation")

    def get(self, key, value):
        return self._parameter_dict.get(key, value)

    def update(self, parameter_dict):
        self._parameter_dict.update(parameter_dict)
        self.__dict__.update(self._parameter_dict)

    def to_dict(self):
        return self._parameter_dict

    @classmethod
    def from_yaml(cls, config_file_path):
        with open(config_file_path) as f:
            return cls(yaml.safe_load(f))
