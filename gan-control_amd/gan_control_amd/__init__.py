"""gan_control_amd -- MI355X-native StyleGAN2 training hot path for gan-control.

Layout mirrors the reference's ``src/gan_control`` for the files on the path:
``models/op`` is the operator socket the reference leaves open (gan_model.py:19-50,
non_leaking.py:6), ``models/gan_model.py`` the Generator/Discriminator with the reference's
constructor arguments and state_dict keys, ``trainers/`` the data-parallel G+D step.
"""
__version__ = '0.1.0'
