"""Policy registry with habitat-baselines' `baseline_registry` surface
(`register_policy` / `get_policy`), so trainers construct the policy as
`baseline_registry.get_policy(config.MODEL.policy_name).from_config(config,
observation_space, action_space)` (base_il_trainer.py:61-66,
ddppo_waypoint_trainer.py:110-115).  When habitat-baselines is importable the
classes are ALSO registered there, replacing the reference's own."""


class _Registry:
    _policies = {}

    @classmethod
    def register_policy(cls, to_register=None, *, name=None):
        def wrap(c):
            cls._policies[name or c.__name__] = c
            try:  # drop-in: make habitat's trainers find the HIP policies
                from habitat_baselines.common.baseline_registry import baseline_registry as hb

                hb.register_policy(c, name=name or c.__name__)
            except Exception:
                pass
            return c

        return wrap if to_register is None else wrap(to_register)

    @classmethod
    def get_policy(cls, name):
        return cls._policies.get(name)


baseline_registry = _Registry


def build_model(config, observation_space, action_space):
    """BASELINE.json's `build_model`: alias of the registry construction path."""
    cls = baseline_registry.get_policy(config.MODEL.policy_name)
    if cls is None:
        raise KeyError(f"unknown policy {config.MODEL.policy_name!r}")
    return cls.from_config(config=config, observation_space=observation_space,
                           action_space=action_space)
