function X = mc_svt(OH, Omega, Imax, tau, rho)
% Drop-in for benchmark_algorithms/mc_svt.m.
  X = jstsp_mex('mc_svt', OH, Omega, Imax, tau, rho);
end
