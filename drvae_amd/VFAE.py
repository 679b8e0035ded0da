"""Variational Fair Autoencoder / SSVAE -- counterpart of reference ``src/VFAE.py``."""
from ._model_base import ELBOModel


class VFAE(ELBOModel):
    """p(x,z1,z2,y) = p(z2)p(y)p(z1|z2,y)p(x|z1); q(z1|x) q(y|z1) q(z2|z1,y)  (arXiv:1511.00830)."""
    kind = 'vfae'
    fit_patience = 40       # src/VFAE.py:536

    def __init__(self, dim_x, dim_s, dim_y, dim_h_en_z1=(50, 50), dim_h_de_z1=(50, 50), dim_h_en_z2=(50, 50),
                 dim_h_de_x=(50, 50), dim_h_clf=(50, 50), dim_z1=50, dim_z2=50, type_rec='binary',
                 type_y='discrete', prior_y='uniform', semi_supervised=False, clf_1sig=False, epochs=500,
                 batch_size=100, nonlinearity='softplus', learning_rate=0.001, optim_alg='adam', L=1,
                 weight_decay=None, dropout_rate=0., input_x_dropout=0., add_noise_var=0., yloss_rate=1.,
                 anneal_yloss_offset=0, use_MMD=True, kernel_MMD='rbf_fourier', mmd_rate=1., use_s=False,
                 random_seed=12345, log_txt=None, weight_norm=False, device=None):
        super().__init__()
        args = dict(locals())
        args.pop('self')
        args.pop('__class__', None)
        self._init_common(args)

    def loss_function(self, x1, s, y, has_y, noise=None):
        self._warn_empty_groups(has_y * 0, has_y)
        return super().loss_function(noise=noise, x1=x1, s=s, y=y, has_y=has_y)

    def evaluate_performance(self, x1, s, y, has_y, return_full_data=False):
        """(perf dict, summary string) of src/VFAE.py:472-521"""
        return self._evaluate(x1, None, s, y, None, has_y, return_full_data)
