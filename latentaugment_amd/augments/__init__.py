"""Plugin registry with the reference's interface (augments/__init__.py:28-72 of the reference):

    from latentaugment_amd.augments import create_augment
    augment = create_augment(opt)          # opt.aug == 'latent'
    augment.set_input(data); augment.forward(); out = augment.get_output()
"""
import importlib

from .base_aug import BaseAugment


def find_augment_using_name(augment_name):
    """Import "<this package>.<augment_name>_aug" and return the BaseAugment subclass named <AugmentName>Augment
    (case-insensitive), as the reference does for "augments.<name>_aug"."""
    module = importlib.import_module(f'{__name__}.{augment_name}_aug')
    target = augment_name.replace('_', '') + 'augment'
    augment = None
    for name, cls in module.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseAugment):
            augment = cls
    if augment is None:
        raise ImportError('In %s_aug.py, there should be a subclass of BaseAugment with class name that matches %s in '
                          'lowercase.' % (augment_name, target))
    return augment


def get_option_setter(augment_name):
    """Return the static method <modify_commandline_options> of the augment class."""
    return find_augment_using_name(augment_name).modify_commandline_options


def create_augment(opt):
    """Create an augment pipeline given the options (opt.aug selects the class)."""
    instance = find_augment_using_name(opt.aug)(opt)
    print('Augment [%s] was created' % type(instance).__name__)
    return instance
