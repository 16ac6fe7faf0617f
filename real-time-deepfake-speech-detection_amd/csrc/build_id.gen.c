const char afx_build_id_str[] = "712918d965bc";
