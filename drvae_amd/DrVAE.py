"""Drug Response VAE -- counterpart of reference ``src/DrVAE.py`` (class ``DrVAE``): same
constructor arguments, sub-module names, ``state_dict`` keys, ``forward`` /
``loss_function`` / ``run_on_batch`` entry points; the ELBO train step runs as the fused
HIP launch sequence of ``drvae_amd.engine`` (see ``_model_base.ELBOModel``)."""
from ._model_base import ELBOModel


class DrVAE(ELBOModel):
    """p(x1,x2,z1,z2,z3,y) = p(z3)p(y)p(z1|z3,y)p(z2|z1)p(x1|z1)p(x2|z2) with posteriors
    q(z1|x1), q(z2|x2) (shared encoder), q(y|z1,z2), q(z3|z1,y)  (arXiv:1706.08203)."""
    kind = 'drvae'

    def __init__(self, dim_x, dim_s, dim_y, dim_c=1, dim_m=1, dim_h_en_z1=(50, 50), dim_h_de_z1=(50, 50),
                 dim_h_en_z2Fz1=(50), dim_h_en_z3=(50, 50), dim_h_de_x=(50, 50), dim_h_clf=(50, 50), dim_z1=50,
                 dim_z3=50, type_rec='binary', clf_z1z2=True, type_y='discrete', prior_y='uniform',
                 clf_1sig=False, epochs=500, batch_size=100, nonlinearity='softplus', learning_rate=0.001,
                 optim_alg='adam', L=1, weight_decay=None, dropout_rate=0., input_x_dropout=0., add_noise_var=0.,
                 yloss_rate=1., anneal_yloss_offset=0, use_MMD=True, kernel_MMD='rbf_fourier', mmd_rate=1.,
                 kl_qz2pz2_rate=1., pertloss_rate=0.1, anneal_perturb_rate_itermax=1,
                 anneal_perturb_rate_offset=0, use_s=False, use_c=False, use_m=False, random_seed=12345,
                 log_txt=None, weight_norm=False, device=None):
        super().__init__()
        args = dict(locals())
        args.pop('self')
        args.pop('__class__', None)
        self._init_common(args)

    def loss_function(self, x1, x2, s, y, has_x2, has_y, noise=None):
        self._warn_empty_groups(has_x2, has_y)
        return super().loss_function(noise=noise, x1=x1, x2=x2, s=s, y=y, has_x2=has_x2, has_y=has_y)

    def evaluate_performance(self, x1, x2, s, y, has_x2, has_y, return_full_data=False):
        """(perf dict, summary string) of src/DrVAE.py:640-741"""
        return self._evaluate(x1, x2, s, y, has_x2, has_y, return_full_data)
