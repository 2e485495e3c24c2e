"""Name -> class registry mirroring the part of the reference's `registry` the hot path uses
(minigpt4/common/registry.py:83-109 `register_model`, :260-262 `get_model_class`)."""


class _Registry:
    def __init__(self):
        self.mapping = {"model_name_mapping": {}, "lr_scheduler_name_mapping": {}}

    def register_model(self, name):
        def wrap(cls):
            if name in self.mapping["model_name_mapping"] and self.mapping["model_name_mapping"][name] is not cls:
                raise KeyError(f"Name '{name}' already registered for {self.mapping['model_name_mapping'][name]}.")
            self.mapping["model_name_mapping"][name] = cls
            return cls
        return wrap

    def register_lr_scheduler(self, name):
        def wrap(cls):
            self.mapping["lr_scheduler_name_mapping"][name] = cls
            return cls
        return wrap

    def get_model_class(self, name):
        return self.mapping["model_name_mapping"].get(name, None)

    def get_lr_scheduler_class(self, name):
        return self.mapping["lr_scheduler_name_mapping"].get(name, None)

    def list_models(self):
        return sorted(self.mapping["model_name_mapping"].keys())


registry = _Registry()
