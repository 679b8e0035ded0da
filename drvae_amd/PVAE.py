"""Drug Perturbation VAE -- counterpart of reference ``src/PVAE.py`` (class ``PVAE``)."""
from ._model_base import ELBOModel


class PVAE(ELBOModel):
    """p(x1,x2,z1,z2) = p(z1)p(z2|z1)p(x1|z1)p(x2|z2); q(z1|x1), q(z2|x2) share one encoder."""
    kind = 'pvae'
    prior_y = None          # read by the reference ctor (src/PVAE.py:77) although PVAE has no y

    def __init__(self, dim_x, dim_s, dim_y, dim_c=1, dim_m=1, dim_h_en_z1=(50, 50), dim_h_en_z2Fz1=(50),
                 dim_h_de_x=(50, 50), dim_z1=50, type_rec='binary', epochs=500, batch_size=100,
                 nonlinearity='softplus', learning_rate=0.001, optim_alg='adam', L=1, weight_decay=None,
                 dropout_rate=0., input_x_dropout=0., add_noise_var=0., use_MMD=True, kernel_MMD='rbf_fourier',
                 mmd_rate=1., kl_qz2pz2_rate=1., pertloss_rate=0.1, anneal_perturb_rate_itermax=1,
                 anneal_perturb_rate_offset=0, use_s=False, use_c=False, use_m=False, random_seed=12345,
                 log_txt=None, weight_norm=False, device=None):
        super().__init__()
        args = dict(locals())
        args.pop('self')
        args.pop('__class__', None)
        self._init_common(args)

    def loss_function(self, x1, x2, s, has_x2, noise=None):
        self._warn_empty_groups(has_x2, has_x2 * 0)
        return super().loss_function(noise=noise, x1=x1, x2=x2, s=s, has_x2=has_x2)

    def evaluate_performance(self, x1, x2, s, has_x2, return_full_data=False):
        """(perf dict, summary string) of src/PVAE.py:479-552"""
        return self._evaluate(x1, x2, s, None, has_x2, None, return_full_data)
