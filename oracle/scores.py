"""ORACLE -- test infrastructure, not product code (see oracle/gp.py for the rules and the "parity unpinned"
statement).  CPU restatement of src/scorefunctions.jl:6-16, line by line."""
import numpy as np


def se(y_true, y_pred):
    return (np.asarray(y_true) - np.asarray(y_pred)) ** 2                 # :6


def mse(y_true, y_pred):
    return float(np.mean(se(y_true, y_pred)))                             # :7


def sse(y_true, y_pred):
    v = se(y_true, y_pred)
    return float(np.std(v, ddof=1) / np.sqrt(v.shape[0]))                 # :8  (Statistics.std: corrected)


def ae(y_true, y_pred):
    return np.abs(np.asarray(y_true) - np.asarray(y_pred))                # :11


def mae(y_true, y_pred):
    return float(np.mean(ae(y_true, y_pred)))                             # :12


def sae(y_true, y_pred):
    v = ae(y_true, y_pred)
    return float(np.std(v, ddof=1) / np.sqrt(v.shape[0]))                 # :13


def nlpd(y_true, mu, var):
    """-mean(logpdf(Normal(mu_i, sqrt(var_i)), y_i))  (:16); Distributions' normal logpdf is
    -(z^2 + log 2pi)/2 - log(sigma) with z = (y - mu)/sigma."""
    y_true, mu, var = map(np.asarray, (y_true, mu, var))
    sd = np.sqrt(var)
    z = (y_true - mu) / sd
    return float(-np.mean(-(z * z + np.log(2 * np.pi)) / 2 - np.log(sd)))
