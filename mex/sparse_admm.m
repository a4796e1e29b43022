function [S, convergence_error] = sparse_admm(Htrue, OH, Dr, Dt, Imax)
% Drop-in for benchmark_algorithms/sparse_admm.m.
  [S, convergence_error] = jstsp_mex('sparse_admm', Htrue, OH, Dr, Dt, Imax);
end
