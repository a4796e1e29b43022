function [X, convergence_error] = mc_admm(Htrue, OH, Omega, Imax, tau, rho)
% Drop-in for benchmark_algorithms/mc_admm.m.
  [X, convergence_error] = jstsp_mex('mc_admm', Htrue, OH, Omega, Imax, tau, rho);
end
