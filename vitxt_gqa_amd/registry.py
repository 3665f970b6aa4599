"""Host-side mirror of the slice of ``pythia/common/registry.py`` the T2S path touches
(``register_model`` :159-185, ``register_loss``, ``register_metric`` :101-127, ``register``/``get`` :232-321,
``get_model_class``, ``get_loss_class``, ``get_metric_class``).  Same names and argument meaning, so the model file reads like the reference's
and can be pointed at the reference's own registry instead."""


class Registry:
    mapping = {"model_name_mapping": {}, "loss_name_mapping": {}, "metric_name_mapping": {}, "optimizer_name_mapping": {}, "state": {}}

    @classmethod
    def register_model(cls, name):
        def wrap(model_cls):
            from .base_model import BaseModel
            assert issubclass(model_cls, BaseModel), "All models must inherit BaseModel class"
            cls.mapping["model_name_mapping"][name] = model_cls
            return model_cls
        return wrap

    @classmethod
    def register_loss(cls, name):
        def wrap(loss_cls):
            cls.mapping["loss_name_mapping"][name] = loss_cls
            return loss_cls
        return wrap

    @classmethod
    def register_metric(cls, name):
        def wrap(metric_cls):
            cls.mapping["metric_name_mapping"][name] = metric_cls
            return metric_cls
        return wrap

    @classmethod
    def register(cls, name, obj):
        path = name.split(".")
        cur = cls.mapping["state"]
        for part in path[:-1]:
            cur = cur.setdefault(part, {})
        cur[path[-1]] = obj

    @classmethod
    def get(cls, name, default=None, no_warning=False):
        value = cls.mapping["state"]
        for sub in name.split("."):
            if not isinstance(value, dict):
                return default
            value = value.get(sub, default)
            if value is default:
                break
        return value

    @classmethod
    def unregister(cls, name):
        return cls.mapping["state"].pop(name, None)

    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model_name_mapping"].get(name, None)

    @classmethod
    def get_loss_class(cls, name):
        return cls.mapping["loss_name_mapping"].get(name, None)

    @classmethod
    def get_metric_class(cls, name):
        return cls.mapping["metric_name_mapping"].get(name, None)

    @classmethod
    def get_optimizer_class(cls, name):
        return cls.mapping["optimizer_name_mapping"].get(name, None)


registry = Registry()
