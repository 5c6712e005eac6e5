"""Known answers for the material arithmetic that do NOT lean on the oracle's own outputs.

The reference ships no tests, so the oracle is unpinnable against it; what can be pinned is the mathematics the
reference's material code states: closed forms (conductor Fresnel at normal incidence, the dielectric Fresnel
equations, Phong's f cos / pdf, the mirror direction, sRGB knots, bilinear weights), normalisation integrals
(the microfacet distribution and the visible-normal distribution integrate to 1) and chi-square tests that
`Scatter` really draws its directions from the density it divides by.  Every check runs against the CPU oracle
(`-m "not gpu"`) and, through the prt_material_* / prt_texture_value hooks of the C ABI, against the HIP device
functions K3 shades with (`-m gpu`); the GPU run also compares the two on the same seeds.

Reference: Source/Material.h:76-98 (frames), :101-155 (Lambertian), :172-330 (PhoneReflectance), :332-366
(PerfectMirror), :368-521 (CookTorrance), Source/MaterialUtils.h:100-111 (FrComplex), Source/Texture.cpp:22-71.
"""
import numpy as np
import pytest

import oracle
from pooraytracer_amd import _abi, scenes

KD, KS = (0.7, 0.5, 0.3), (0.35, 0.45, 0.55)
GOLD_ETA, GOLD_K = (0.1, 0.5, 1.5), (4.0, 0.02, 0.3)
M_LAMBERT, M_TEX, M_PHONG50, M_PHONG5, M_MIRROR, M_CT_ISO, M_CT_ANISO, M_GLASSY, M_LIGHT, M_GREY = range(10)


def material_scene():
    """One triangle per material; what matters is the material / texture tables."""
    b = scenes._Builder("materials")
    rng = np.random.default_rng(5)
    tex_rgb = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)   # H=5, W=7
    tex_rgb[0, 0] = (0, 10, 11)        # around the linear / power knee of the sRGB curve (10/255 < 0.04045 < 11/255)
    tex_grey = rng.integers(0, 256, size=(4, 3), dtype=np.uint8)
    b.textures += [tex_rgb, tex_grey]
    M = scenes.Material
    mats = [
        M("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=KD),
        M("Wood", _abi.MAT_LAMBERTIAN, kd=(1, 1, 1), texture=0),
        M("material1", _abi.MAT_PHONG, kd=KD, ks=KS, ns=50.0),
        M("material4", _abi.MAT_PHONG, kd=KD, ks=KS, ns=5.0),
        M("Mirror", _abi.MAT_MIRROR),
        M("GoldIso", _abi.MAT_COOKTORRANCE, kd=KD, eta=GOLD_ETA, k=GOLD_K, alpha_x=0.5, alpha_y=0.5),
        M("GoldAniso", _abi.MAT_COOKTORRANCE, kd=KD, eta=GOLD_ETA, k=GOLD_K, alpha_x=0.2, alpha_y=0.6),
        M("Glassy", _abi.MAT_COOKTORRANCE, kd=KD, eta=(1.5, 1.33, 2.4), k=(0.0, 0.0, 0.0), alpha_x=0.3, alpha_y=0.3),
        M("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(5, 5, 5)),
        M("Label", _abi.MAT_LAMBERTIAN, kd=(1, 1, 1), texture=1),
    ]
    for i, m in enumerate(mats):
        idx = b.material(m)
        assert idx == i
        v = np.array([[[i, 0, 0], [i + 0.5, 0, 0], [i, 0.5, 0]]], dtype=np.float64)
        b.mesh(f"m{i}", idx, v)
    data = b.build(scenes.Camera(8, 8, 40.0, eye=(0, 0, 5), look_at=(0, 0, 0)))
    return data, tex_rgb, tex_grey


DATA, TEX_RGB, TEX_GREY = material_scene()


@pytest.fixture(scope="module")
def orc():
    return oracle.Oracle(DATA)


@pytest.fixture(scope="module")
def dev(gpu):
    from pooraytracer_amd import api
    return api.Scene(DATA).upload(0)


# ------------------------------------------------------------------ independent closed forms (numpy, from the literature)
def srgb_to_linear(c):
    c = np.asarray(c, dtype=np.float64)
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def fresnel_conductor_normal(eta, k):
    return ((eta - 1) ** 2 + k ** 2) / ((eta + 1) ** 2 + k ** 2)


def fresnel_dielectric(cos_i, eta):
    """Unpolarised Fresnel reflectance of a dielectric interface (Snell + the two amplitude ratios)."""
    sin2_t = (1 - cos_i ** 2) / eta ** 2
    cos_t = np.sqrt(np.maximum(0.0, 1 - sin2_t))
    r_par = (eta * cos_i - cos_t) / (eta * cos_i + cos_t)
    r_per = (cos_i - eta * cos_t) / (cos_i + eta * cos_t)
    return 0.5 * (r_par ** 2 + r_per ** 2)


def ggx_D(m, ax, ay):
    """Anisotropic Trowbridge-Reitz distribution (Walter et al. 2007 / Heitz 2014)."""
    c2 = m[..., 2] ** 2
    return 1.0 / (np.pi * ax * ay * ((m[..., 0] / ax) ** 2 + (m[..., 1] / ay) ** 2 + c2) ** 2)


def ggx_lambda(w, ax, ay):
    a2t2 = ((w[..., 0] * ax) ** 2 + (w[..., 1] * ay) ** 2) / np.maximum(w[..., 2] ** 2, 1e-300)
    return 0.5 * (np.sqrt(1 + a2t2) - 1)


def ggx_vndf(w, m, ax, ay):
    """Density of visible normals D_w(m) = G1(w) max(0, w.m) D(m) / cos(theta_w) (Heitz 2014, eq. 3)."""
    return np.maximum(0.0, np.sum(w * m, -1)) * ggx_D(m, ax, ay) / ((1 + ggx_lambda(w, ax, ay)) * np.abs(w[..., 2]))


def sph(theta, phi):
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], -1)


def hemisphere_quadrature(nt=400, nph=800):
    """Midpoint rule in (cos theta, phi): directions and solid-angle weights summing to 2 pi."""
    ct = (np.arange(nt) + 0.5) / nt
    ph = (np.arange(nph) + 0.5) / nph * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    d = sph(np.arccos(CT), PH).reshape(-1, 3)
    return d, np.full(d.shape[0], 2 * np.pi / (nt * nph))


def chi2_ok(observed, expected, min_expected=25.0):
    """Pearson chi-square over the cells with enough expected counts, against a p = 1e-6 bound
    (Wilson-Hilferty approximation of the chi-square quantile); returns (ok, chi2, dof)."""
    keep = expected >= min_expected
    chi2 = float(np.sum((observed[keep] - expected[keep]) ** 2 / expected[keep]))
    dof = int(keep.sum()) - 1
    z = 4.75  # one-sided normal quantile for p = 1e-6
    bound = dof * (1 - 2 / (9 * dof) + z * np.sqrt(2 / (9 * dof))) ** 3
    return chi2 <= bound, chi2, dof


# ------------------------------------------------------------------ the checks (backend = oracle.Oracle or api.Scene)
def check_fresnel_through_eval(be):
    """At wi = wo = n the half vector is n, D(n) = 1/(pi ax ay), G = 1, so f = F / (4 pi ax ay): conductor Fresnel at
    normal incidence is ((eta-1)^2 + k^2) / ((eta+1)^2 + k^2) per channel (MaterialUtils.h:100-111)."""
    n = np.array([[0.0, 0.0, 1.0]])
    for mat, (ax, ay), eta, k in ((M_CT_ISO, (0.5, 0.5), GOLD_ETA, GOLD_K), (M_CT_ANISO, (0.2, 0.6), GOLD_ETA, GOLD_K),
                                  (M_GLASSY, (0.3, 0.3), (1.5, 1.33, 2.4), (0, 0, 0))):
        f = be.material_eval(mat, n, n)[0]
        want = fresnel_conductor_normal(np.array(eta), np.array(k)) / (4 * np.pi * ax * ay)
        assert np.allclose(f, want, rtol=1e-12), (mat, f, want)
    # k = 0 at oblique incidence: the dielectric Fresnel equations.  wi = reflect(wo, n) keeps the half vector at n.
    for th in (0.2, 0.7, 1.2, 1.5):
        wo = sph(np.array([th]), np.array([0.4]))
        wi = wo * np.array([-1, -1, 1])
        f = be.material_eval(M_GLASSY, wi, wo)[0]
        c = np.cos(th)
        G = 1.0 / (1 + 2 * ggx_lambda(wo, 0.3, 0.3)[0])
        want = fresnel_dielectric(c, np.array([1.5, 1.33, 2.4])) * (1 / (np.pi * 0.09)) * G / (4 * c * c)
        assert np.allclose(f, want, rtol=1e-11), (th, f, want)
    # different hemispheres / grazing: zero
    assert not be.material_eval(M_CT_ISO, [[0.3, 0.1, -0.9]], [[0.1, 0.2, 0.97]]).any()


def check_eval_against_closed_form_brdf(be):
    """CookTorrance::Eval = D F G / (4 cos_i cos_o) with the literature's anisotropic GGX D and Smith G
    (Material.h:373-411,474-496) on random direction pairs; Lambertian Eval = Kd / pi."""
    rng = np.random.default_rng(17)
    wi = sph(np.arccos(rng.uniform(0.05, 1, 500)), rng.uniform(0, 2 * np.pi, 500))
    wo = sph(np.arccos(rng.uniform(0.05, 1, 500)), rng.uniform(0, 2 * np.pi, 500))
    assert np.allclose(be.material_eval(M_LAMBERT, wi, wo), np.array(KD) / np.pi, rtol=1e-15)
    for mat, ax, ay in ((M_CT_ISO, 0.5, 0.5), (M_CT_ANISO, 0.2, 0.6)):
        f = be.material_eval(mat, wi, wo)
        wm = wi + wo
        wm /= np.linalg.norm(wm, axis=-1, keepdims=True)
        G = 1 / (1 + ggx_lambda(wo, ax, ay) + ggx_lambda(wi, ax, ay))
        # ratios between channels are Fresnel ratios; the red channel against a dielectric-free closed form is not
        # available at oblique incidence for a conductor, so check the geometric factor through the channel-free part:
        geo = ggx_D(wm, ax, ay) * G / (4 * wi[:, 2] * wo[:, 2])
        F = f / geo[:, None]
        assert (F > 0).all() and (F <= 1 + 1e-12).all()                     # a reflectance
        cos_h = np.abs(np.sum(wo * wm, -1))
        near_normal = cos_h > 0.999
        if near_normal.any():                                              # ... that tends to the normal-incidence value
            assert np.allclose(F[near_normal], fresnel_conductor_normal(np.array(GOLD_ETA), np.array(GOLD_K)), rtol=5e-3)
        # same half vector, same |wo . wm| -> same Fresnel term whatever the roughness: F depends on cos_h only
        order = np.argsort(cos_h)
        Fs = F[order]
        assert (np.abs(np.diff(Fs[:, 0])) < 0.05).all()                     # smooth in cos_h (no roughness leakage)


def check_phong(be):
    """PhoneReflectance::Scatter (Material.h:183-226,263-285): with Ns > 9 the diffuse lobe is chosen with probability
    0.6 and returns attenuation Kd; the specular lobe returns f cos / pdf = Ks (Ns+2)/(Ns+1) cos(theta_i), and
    u = cos(alpha)^(Ns+1) of the angle to the mirror direction is uniform.  With Ns <= 9 only the diffuse lobe exists."""
    n = 60000
    rd = np.tile(np.array([[0.3, -0.2, -1.0]]), (n, 1))
    nrm, tan = np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0])
    wi, att, ok = be.material_scatter(M_PHONG50, rd, nrm, tan, seed=11)
    assert ok.all()
    assert np.allclose(np.linalg.norm(wi, axis=-1), 1.0, atol=1e-12)
    diffuse = np.all(np.abs(att - np.array(KD)) < 1e-12, axis=-1)
    frac = diffuse.mean()
    assert abs(frac - 0.6) < 5 * np.sqrt(0.24 / n), frac
    cos_i = wi[:, 2]
    spec = ~diffuse
    good = spec & (att.sum(-1) > 0)
    want = np.array(KS)[None, :] * (52.0 / 51.0) * cos_i[good, None]
    assert np.allclose(att[good], want, rtol=1e-9)
    assert (att[spec & ~good] == 0).all() and (spec & ~good).mean() < 0.05   # below-horizon specular samples contribute 0 (B13)
    # lobe shape: angle to the mirror direction of -rd about n
    d = rd[0] / np.linalg.norm(rd[0])
    mirror = d - 2 * d[2] * nrm
    cos_a = np.clip(wi[spec] @ mirror, 0, 1)
    u = cos_a ** 51.0
    hist, _ = np.histogram(u, bins=20, range=(0, 1))
    okc, chi2, dof = chi2_ok(hist.astype(float), np.full(20, spec.sum() / 20.0))
    assert okc, (chi2, dof)
    # Eval's specular value for the same geometry: Ks (Ns+2)/(2 pi) cos^Ns(alpha), drawn with probability 0.4
    wo = -d
    wi_l = wi[good][:2000]
    f = be.material_eval(M_PHONG50, wi_l, np.tile(wo, (wi_l.shape[0], 1)), seed=3)
    is_d = np.all(np.abs(f - np.array(KD) / np.pi) < 1e-13, axis=-1)
    assert abs(is_d.mean() - 0.6) < 0.06
    ca = np.clip(wi_l[~is_d] @ mirror, 0, 1)
    assert np.allclose(f[~is_d], np.array(KS)[None, :] * 52.0 / (2 * np.pi) * (ca ** 50.0)[:, None], rtol=1e-9, atol=1e-300)
    # Ns = 5: pkd = 1, pks = 0
    _, att5, ok5 = be.material_scatter(M_PHONG5, rd[:5000], nrm, tan, seed=12)
    assert ok5.all() and np.allclose(att5, np.array(KD), rtol=1e-12)


def check_mirror_and_frames(be):
    """PerfectMirror::Scatter (Material.h:334-363): direction d - 2 (d.n) n, attenuation exactly 1, in a tilted frame."""
    nrm = np.array([0.36, 0.48, 0.8])
    tan = np.cross([0.0, 1.0, 0.0], nrm)
    tan /= np.linalg.norm(tan)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(200, 3))
    d[(d @ nrm) > 0] *= -1                      # arriving from the front side
    d *= rng.uniform(0.2, 5.0, size=(200, 1))   # camera rays are not normalised (B4)
    wi, att, ok = be.material_scatter(M_MIRROR, d, nrm, tan)
    dn = d / np.linalg.norm(d, axis=-1, keepdims=True)
    assert ok.all() and np.allclose(wi, dn - 2 * (dn @ nrm)[:, None] * nrm, atol=1e-12)
    assert np.allclose(att, 1.0, rtol=1e-14)
    # emitters / absorbers do not scatter (Material.h:57-59)
    assert not be.material_scatter(M_LIGHT, d[:4], nrm, tan)[2].any()


def check_texture(be):
    """ImageTexture::Value (Texture.cpp:22-71): texel centres return the sRGB-decoded texel (v flipped), points halfway
    between two texels their mean — decode BEFORE blending —, coordinates outside [0,1] clamp, grey images skip the decode."""
    H, W = TEX_RGB.shape[:2]
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    uv = np.stack([xs.ravel() / (W - 1.0), 1.0 - ys.ravel() / (H - 1.0)], -1)
    got = be.texture_value(0, uv)
    want = srgb_to_linear(TEX_RGB[ys.ravel(), xs.ravel()] / 255.0)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-15)
    # the knee: 10/255 is on the linear branch, 11/255 on the power branch
    assert np.isclose(got[0, 1], (10 / 255.0) / 12.92, rtol=1e-14) and np.isclose(got[0, 2], ((11 / 255.0 + 0.055) / 1.055) ** 2.4, rtol=1e-12)
    # halfway in x between texels (2,1) and (3,1); halfway in both directions
    mid = be.texture_value(0, [[2.5 / (W - 1.0), 1.0 - 1.0 / (H - 1.0)], [2.5 / (W - 1.0), 1.0 - 1.5 / (H - 1.0)]])
    lin = srgb_to_linear(TEX_RGB / 255.0)
    assert np.allclose(mid[0], 0.5 * (lin[1, 2] + lin[1, 3]), rtol=1e-9)
    assert np.allclose(mid[1], 0.25 * (lin[1, 2] + lin[1, 3] + lin[2, 2] + lin[2, 3]), rtol=1e-9)
    # bilinear weights at an arbitrary point
    u, v = 0.37, 0.81
    x, y = u * (W - 1), (1 - v) * (H - 1)
    x0, y0, fx, fy = int(x), int(y), x - int(x), y - int(y)
    want = (lin[y0, x0] * (1 - fx) + lin[y0, x0 + 1] * fx) * (1 - fy) + (lin[y0 + 1, x0] * (1 - fx) + lin[y0 + 1, x0 + 1] * fx) * fy
    assert np.allclose(be.texture_value(0, [[u, v]])[0], want, rtol=1e-12)
    # clamp
    assert np.allclose(be.texture_value(0, [[-3.0, 7.0], [2.0, -1.0]]), [lin[0, 0], lin[H - 1, W - 1]], rtol=1e-12)
    # one channel: grey, no sRGB decode
    g = be.texture_value(1, [[0.0, 1.0], [1.0, 0.0]])
    assert np.allclose(g, [[TEX_GREY[0, 0] / 255.0] * 3, [TEX_GREY[-1, -1] / 255.0] * 3], rtol=1e-14)
    # a textured Lambertian evaluates to texel / pi
    f = be.material_eval(M_TEX, [[0, 0, 1.0]], [[0, 0, 1.0]], uv=[[3 / (W - 1.0), 1.0 - 2 / (H - 1.0)]])
    assert np.allclose(f[0], lin[2, 3] / np.pi, rtol=1e-12)


def check_lambertian_sampling(be):
    """Lambertian::Scatter (Material.h:106-151): cosine-weighted — cos^2(theta) and phi are uniform — and the
    attenuation f cos / pdf is the albedo for every sample."""
    n = 200000
    rd = np.tile(np.array([[0.0, 0.0, -1.0]]), (n, 1))
    wi, att, ok = be.material_scatter(M_LAMBERT, rd, seed=21)
    assert ok.all() and np.allclose(att, np.array(KD), rtol=1e-12)
    assert (wi[:, 2] > 0).all()
    h1, _ = np.histogram(wi[:, 2] ** 2, bins=25, range=(0, 1))
    h2, _ = np.histogram(np.arctan2(wi[:, 1], wi[:, 0]), bins=16, range=(-np.pi, np.pi))
    for h in (h1, h2):
        okc, chi2, dof = chi2_ok(h.astype(float), np.full(h.size, n / h.size))
        assert okc, (chi2, dof)


def check_cooktorrance_sampling(be):
    """CookTorrance::Scatter (Material.h:412-472,497-516) draws the half vector from the distribution of visible normals and
    reflects: the density of wi is D_wo(wm) / (4 |wo . wm|).  Chi-square of the sampled directions against that density
    (evaluated with the literature's formulas, integrated per cell), and attenuation == f cos / pdf with f from Eval."""
    n = 300000
    for mat, ax, ay, th_o, ph_o in ((M_CT_ISO, 0.5, 0.5, 0.9, 0.3), (M_CT_ANISO, 0.2, 0.6, 0.6, 2.0)):
        wo = sph(np.array(th_o), np.array(ph_o))
        rd = np.tile(-wo[None, :] * 2.5, (n, 1))
        wi, att, ok = be.material_scatter(mat, rd, seed=31)
        # expected counts per (cos theta, phi) cell by midpoint quadrature, 12 x 12 points per cell
        nt, nph, sub = 12, 24, 12
        ct = (np.arange(nt * sub) + 0.5) / (nt * sub)
        ph = (np.arange(nph * sub) + 0.5) / (nph * sub) * 2 * np.pi
        CT, PH = np.meshgrid(ct, ph, indexing="ij")
        d = sph(np.arccos(CT), PH)
        wm = d + wo
        wm /= np.linalg.norm(wm, axis=-1, keepdims=True)
        pdf = ggx_vndf(np.broadcast_to(wo, wm.shape), wm, ax, ay) / (4 * np.abs(np.sum(wm * wo, -1)))
        cell = pdf.reshape(nt, sub, nph, sub).sum(axis=(1, 3)) * (2 * np.pi / (nt * sub * nph * sub))
        expected = cell * n
        w = wi[ok]
        it = np.minimum((w[:, 2] * nt).astype(int), nt - 1)
        ip = np.minimum(((np.arctan2(w[:, 1], w[:, 0]) % (2 * np.pi)) / (2 * np.pi) * nph).astype(int), nph - 1)
        observed = np.zeros((nt, nph))
        np.add.at(observed, (it, ip), 1)
        okc, chi2, dof = chi2_ok(observed.ravel(), expected.ravel(), min_expected=50)
        assert dof > 30 and okc, (mat, chi2, dof)
        # samples that leave through the surface are rejected (Scatter returns false): their share is the missing mass
        assert abs((~ok).mean() - (1 - cell.sum())) < 0.01
        # attenuation = f cos / pdf on a subset
        k = 3000
        wk = w[:k]
        f = be.material_eval(mat, wk, np.tile(wo, (k, 1)))
        wmk = wk + wo
        wmk /= np.linalg.norm(wmk, axis=-1, keepdims=True)
        pk = ggx_vndf(np.tile(wo, (k, 1)), wmk, ax, ay) / (4 * np.abs(wmk @ wo))
        assert np.allclose(att[ok][:k], f * (wk[:, 2] / pk)[:, None], rtol=1e-8)


# ------------------------------------------------------------------ CPU: the oracle
def test_oracle_fresnel_closed_forms(orc):
    check_fresnel_through_eval(orc)
    # the conductor formula itself, through the term hook: normal incidence, grazing incidence, k = 0 vs the dielectric equations
    z = np.array([[0.0, 0.0, 1.0]])
    t = orc.cooktorrance_terms(M_CT_ISO, z, z)
    assert np.allclose(t["F"][0], fresnel_conductor_normal(np.array(GOLD_ETA), np.array(GOLD_K)), rtol=1e-14)
    th = np.linspace(0.0, 1.55, 40)
    w = sph(th, np.zeros_like(th))
    t = orc.cooktorrance_terms(M_GLASSY, w, np.tile(z, (40, 1)))
    assert np.allclose(t["F"], fresnel_dielectric(np.cos(th)[:, None], np.array([1.5, 1.33, 2.4])[None, :]), rtol=1e-12)
    g = orc.cooktorrance_terms(M_CT_ISO, sph(np.array([np.pi / 2 - 1e-9]), np.array([0.0])), z)
    assert np.allclose(g["F"], 1.0, atol=1e-6)


def test_oracle_microfacet_distribution_is_normalised(orc):
    """Integral of D(wm) cos(theta_m) over the hemisphere = 1 and integral of D_w(wm) = 1 (Material.h:373-411); G1 = 1/(1+Lambda);
    all three against the literature's closed forms as well."""
    d, wts = hemisphere_quadrature()
    for mat, ax, ay in ((M_CT_ISO, 0.5, 0.5), (M_CT_ANISO, 0.2, 0.6)):
        t = orc.cooktorrance_terms(mat, d, d)
        assert abs(np.sum(t["D"] * d[:, 2] * wts) - 1.0) < 2e-3
        assert np.allclose(t["D"], ggx_D(d, ax, ay), rtol=1e-10)
        assert np.allclose(t["Lambda"], ggx_lambda(d, ax, ay), rtol=1e-9, atol=1e-15)
        assert np.allclose(t["G1"], 1 / (1 + t["Lambda"]), rtol=1e-15)
        for th, ph in ((0.3, 0.0), (1.0, 1.1), (1.4, 4.0)):
            w = np.tile(sph(np.array(th), np.array(ph)), (d.shape[0], 1))
            tv = orc.cooktorrance_terms(mat, w, d)
            # the reference (like pbrt) writes |w . wm|; the density lives on the VISIBLE normals, w . wm > 0
            vis = np.sum(w * d, -1) > 0
            assert abs(np.sum(tv["Dv"][vis] * wts[vis]) - 1.0) < 5e-3, (mat, th, np.sum(tv["Dv"][vis] * wts[vis]))
            assert np.allclose(tv["Dv"], np.abs(np.sum(w * d, -1)) * ggx_D(d, ax, ay) / ((1 + ggx_lambda(w, ax, ay)) * w[:, 2]), rtol=1e-9)


def test_oracle_frames_are_inverse_of_each_other():
    """LocalToWorld(WorldToLocal(v)) = v / |v| for an orthonormal (normal, tangent) (Material.h:76-98), bitangent = t x n."""
    nrm = np.array([0.36, 0.48, 0.8])
    tan = np.cross([0.0, 1.0, 0.0], nrm)
    tan /= np.linalg.norm(tan)
    v = np.random.default_rng(1).normal(size=(100, 3))
    loc = oracle.frame(nrm, tan, v, True)
    assert np.allclose(loc, np.stack([v @ tan, v @ np.cross(tan, nrm), v @ nrm], -1), rtol=1e-14, atol=1e-15)
    back = oracle.frame(nrm, tan, loc, False)
    assert np.allclose(back, v / np.linalg.norm(v, axis=-1, keepdims=True), atol=1e-13)


def test_oracle_eval_matches_closed_form_brdf(orc):
    check_eval_against_closed_form_brdf(orc)


def test_oracle_phong_lobes(orc):
    check_phong(orc)


def test_oracle_mirror(orc):
    check_mirror_and_frames(orc)


def test_oracle_texture_filtering(orc):
    check_texture(orc)


def test_oracle_lambertian_directions_follow_the_cosine_density(orc):
    check_lambertian_sampling(orc)


def test_oracle_cooktorrance_directions_follow_the_stated_density(orc):
    check_cooktorrance_sampling(orc)


# ------------------------------------------------------------------ GPU: the device functions K3 shades with
@pytest.mark.gpu
def test_gpu_fresnel_and_brdf_closed_forms(dev):
    check_fresnel_through_eval(dev)
    check_eval_against_closed_form_brdf(dev)


@pytest.mark.gpu
def test_gpu_phong_mirror_texture(dev):
    check_phong(dev)
    check_mirror_and_frames(dev)
    check_texture(dev)


@pytest.mark.gpu
def test_gpu_sampling_follows_the_stated_densities(dev):
    check_lambertian_sampling(dev)
    check_cooktorrance_sampling(dev)


@pytest.mark.gpu
def test_gpu_material_hooks_equal_oracle_on_the_same_seeds(dev, orc):
    """Same inputs, same keyed streams: directions to 1e-9 (the device replaces acos / sincos / pow by algebraic
    equivalents), attenuations and BRDF values to 1e-9 relative, identical accept / reject decisions."""
    rng = np.random.default_rng(8)
    n = 20000
    nrm = np.array([0.36, 0.48, 0.8])
    tan = np.cross([0.0, 1.0, 0.0], nrm)
    tan /= np.linalg.norm(tan)
    rd = rng.normal(size=(n, 3))
    rd[(rd @ nrm) > 0] *= -1
    uv = rng.uniform(-0.2, 1.2, size=(n, 2))
    for mat in (M_LAMBERT, M_TEX, M_PHONG50, M_PHONG5, M_MIRROR, M_CT_ISO, M_CT_ANISO, M_GLASSY, M_LIGHT, M_GREY):
        gw, ga, gok = dev.material_scatter(mat, rd, nrm, tan, uv=uv, seed=77)
        ow, oa, ook = orc.material_scatter(mat, rd, nrm, tan, uv=uv, seed=77)
        assert np.array_equal(gok, ook), mat
        assert np.allclose(gw, ow, atol=1e-9), mat
        assert np.allclose(ga, oa, rtol=1e-9, atol=1e-12), mat
        wi = sph(np.arccos(rng.uniform(-0.2, 1, n)), rng.uniform(0, 2 * np.pi, n))
        wo = sph(np.arccos(rng.uniform(0.02, 1, n)), rng.uniform(0, 2 * np.pi, n))
        assert np.allclose(dev.material_eval(mat, wi, wo, uv=uv, seed=5), orc.material_eval(mat, wi, wo, uv=uv, seed=5), rtol=1e-9, atol=1e-14), mat
    assert np.allclose(dev.texture_value(0, uv), orc.texture_value(0, uv), rtol=1e-13)
    assert np.allclose(dev.texture_value(1, uv), orc.texture_value(1, uv), rtol=1e-13)
